// --use-batch-norm (base_network.py:74-79): slim.batch_norm (decay 0.999, center, no scale, epsilon 1e-3) between
// every conv (which then has no bias) and its ReLU + max-pool.  Training mode (base_network.IS_TRAINING fed True by
// the train ops) normalises with the batch moments over (batch, y, x); inference mode uses the moving averages, which
// the reference never updates (no UPDATE_OPS dependency), i.e. mean 0 / variance 1 -- that case is just the fused
// conv kernel with weights scaled by 1/sqrt(1 + eps) and beta as the bias.
//
// Training mode cannot stay fused (the statistics of the conv output are needed before the ReLU): the conv writes its
// plain output z, gather_stats / stats_finalize give (inv, -mean*inv), and the kernels here do the rest:
//   forward : pooled = pool(relu(z*inv + shift + beta)), arg-max code                     (bn_relu_pool_kernel)
//   backward: dy = pooled gradient routed to the arg-max; dbeta = sum dy; with zhat = z*inv + shift
//             dz = inv * (dy - mean(dy) - zhat * mean(dy * zhat))                          (bn_bwd_reduce / bn_bwd_dz)
//             zhat at an arg-max position is (pooled - beta) wherever dy != 0, so both means come from the pooled
//             tensors alone; dz is dense and overwrites z in place, and feeds the dense-dY modes of the conv dW / dX
//             kernels.
#include "common.h"

// CT: compile-time channel count (10 for every layer of this trunk: cheap index arithmetic), 0: use the argument
template <int CT>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const float* __restrict__ z, long z_bstride,
                                                           const float* __restrict__ stat, const float* __restrict__ beta,
                                                           float* __restrict__ pool, long pool_bstride,
                                                           uint8_t* __restrict__ amax, int B, int H, int W, int Carg) {
  const int C = CT ? CT : Carg;
  const int Hp = H >> 1, Wp = W >> 1;
  const unsigned ncell = (unsigned)B * Hp * Wp * C;      // 32-bit index arithmetic (64-bit divisions dominated this kernel)
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < ncell; idx += gridDim.x * 256u) {
    const int o = (int)(idx % (unsigned)C);
    unsigned r = idx / (unsigned)C;
    const int px = (int)(r % (unsigned)Wp); r /= (unsigned)Wp;
    const int py = (int)(r % (unsigned)Hp);
    const int b = (int)(r / (unsigned)Hp);
    const float inv = stat[o], sh = stat[C + o], be = beta[o];
    const float* zp = z + (long)b * z_bstride + ((long)(2 * py) * W + 2 * px) * C + o;
    const float v0 = (zp[0] * inv + sh) + be, v1 = (zp[C] * inv + sh) + be;
    const float v2 = (zp[(long)W * C] * inv + sh) + be, v3 = (zp[(long)W * C + C] * inv + sh) + be;
    float m = v0; int code = 0;
    if (v1 > m) { m = v1; code = 1; }
    if (v2 > m) { m = v2; code = 2; }
    if (v3 > m) { m = v3; code = 3; }
    const long e = ((long)py * Wp + px) * C + o;
    pool[(long)b * pool_bstride + e] = m > 0.f ? m : 0.f;
    amax[(long)b * Hp * Wp * C + e] = (uint8_t)code;
  }
}

int launch_bn_relu_pool(cpp_ctx* ctx, const float* z, long z_bstride, const float* stat, const float* beta, float* pool,
                        long pool_bstride, uint8_t* amax, int B, int H, int W, int C) {
  const long ncell = (long)B * (H / 2) * (W / 2) * C;
  int grid = (int)((ncell + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  prof_begin(ctx);
  if (C == 10) hipLaunchKernelGGL(bn_relu_pool_kernel<10>, dim3(grid), dim3(256), 0, ctx->stream, z, z_bstride, stat, beta, pool,
                                  pool_bstride, amax, B, H, W, C);
  else hipLaunchKernelGGL(bn_relu_pool_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream, z, z_bstride, stat, beta, pool,
                          pool_bstride, amax, B, H, W, C);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// part[blk][2][C] (f64): sum of dy and of dy * zhat over the block's slice of pooled cells; a thread keeps one channel
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dpool, long dpool_bstride,
                                                            const float* __restrict__ pool, long pool_bstride,
                                                            const float* __restrict__ beta, int B, int cells_per_img, int C,
                                                            double* __restrict__ part) {
  __shared__ double r1[256], r2[256];
  const int active = (256 / C) * C;
  const int t = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  if (t < active) {
    const float be = beta[t % C];
    const long ncell = (long)B * cells_per_img;
    for (unsigned idx = blockIdx.x * (unsigned)active + t; idx < (unsigned)ncell; idx += gridDim.x * (unsigned)active) {
      const int b = (int)(idx / (unsigned)cells_per_img);
      const long e = (long)idx - (long)b * cells_per_img;
      const float pv = pool[(long)b * pool_bstride + e];
      if (pv > 0.f) {
        const float g = dpool[(long)b * dpool_bstride + e];
        s1 += (double)g;
        s2 += (double)g * (double)(pv - be);
      }
    }
  }
  r1[t] = s1; r2[t] = s2;
  __syncthreads();
  if (t < C) {
    double a1 = 0.0, a2 = 0.0;
    for (int k = t; k < active; k += C) { a1 += r1[k]; a2 += r2[k]; }
    part[((long)blockIdx.x * 2 + 0) * C + t] = a1;
    part[((long)blockIdx.x * 2 + 1) * C + t] = a2;
  }
}

// means[2][C] = (sum dy, sum dy*zhat) / N; dbeta = sum dy.  One 64-lane wave per channel: lanes stride over the block
// partials, fixed-order butterfly combine (deterministic; a single thread walking 256 partials took 60 us)
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nblk, int C, double n,
                                                             float* __restrict__ means, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a1 += part[((long)k * 2 + 0) * C + c]; a2 += part[((long)k * 2 + 1) * C + c]; }
  for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
  if (threadIdx.x == 0) {
    means[c] = (float)(a1 / n);
    means[C + c] = (float)(a2 / n);
    dbeta[c] = (float)a1;
  }
}

// z (plain conv output) -> dz in place
template <int CT>
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(float* __restrict__ z, long z_bstride, const float* __restrict__ stat,
                                                        const float* __restrict__ means, const float* __restrict__ dpool,
                                                        long dpool_bstride, const float* __restrict__ pool, long pool_bstride,
                                                        const uint8_t* __restrict__ amax, int B, int H, int W, int Carg) {
  const int C = CT ? CT : Carg;
  const int Hp = H >> 1, Wp = W >> 1;
  const unsigned n = (unsigned)B * H * W * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
    const int o = (int)(idx % (unsigned)C);
    unsigned r = idx / (unsigned)C;
    const int x = (int)(r % (unsigned)W); r /= (unsigned)W;
    const int y = (int)(r % (unsigned)H);
    const int b = (int)(r / (unsigned)H);
    float* zp = z + (long)b * z_bstride + ((long)y * W + x) * C + o;
    const float inv = stat[o];
    const float zhat = *zp * inv + stat[C + o];
    float dy = 0.f;
    const int py = y >> 1, px = x >> 1;
    if (py < Hp && px < Wp) {
      const long e = ((long)py * Wp + px) * C + o;
      if (pool[(long)b * pool_bstride + e] > 0.f && amax[(long)b * Hp * Wp * C + e] == (uint8_t)((y & 1) * 2 + (x & 1)))
        dy = dpool[(long)b * dpool_bstride + e];
    }
    *zp = inv * ((dy - means[o]) - zhat * means[C + o]);
  }
}

#define BN_BWD_BLOCKS 256
size_t bn_bwd_part_doubles(int C) { return (size_t)BN_BWD_BLOCKS * 2 * C; }

int launch_bn_backward(cpp_ctx* ctx, float* z, long z_bstride, const float* stat, const float* beta, const float* dpool,
                       long dpool_bstride, const float* pool, long pool_bstride, const uint8_t* amax, int B, int H, int W,
                       int C, double* part, float* means, float* dbeta) {
  const int cells = (H / 2) * (W / 2) * C;
  prof_begin(ctx);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(BN_BWD_BLOCKS), dim3(256), 0, ctx->stream, dpool, dpool_bstride, pool,
                     pool_bstride, beta, B, cells, C, part);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(64), 0, ctx->stream, part, BN_BWD_BLOCKS, C,
                     (double)B * H * W, means, dbeta);
  const long n = (long)B * H * W * C;
  int grid = (int)((n + 255) / 256);
  if (grid > 8192) grid = 8192;
  if (C == 10) hipLaunchKernelGGL(bn_bwd_dz_kernel<10>, dim3(grid), dim3(256), 0, ctx->stream, z, z_bstride, stat, means, dpool,
                                  dpool_bstride, pool, pool_bstride, amax, B, H, W, C);
  else hipLaunchKernelGGL(bn_bwd_dz_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream, z, z_bstride, stat, means, dpool,
                          dpool_bstride, pool, pool_bstride, amax, B, H, W, C);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// --use-batch-norm (base_network.py:74-79): slim.batch_norm (decay 0.999, center, no scale, epsilon 1e-3) between
// every conv (which then has no bias) and its ReLU + max-pool.  Training mode (base_network.IS_TRAINING fed True by
// the train ops) normalises with the batch moments over (batch, y, x); inference mode uses the moving averages, which
// the reference never updates (no UPDATE_OPS dependency), i.e. mean 0 / variance 1 -- that case is just the fused
// conv kernel with weights scaled by 1/sqrt(1 + eps) and beta as the bias.
//
// Training mode cannot stay fused (the statistics of the conv output are needed before the ReLU): the conv writes its
// plain output z and the kernels here do the rest, for up to four same-shaped networks per launch (blockIdx.y):
//   forward : (inv, -mean*inv) per channel from block partials (f64, fixed-order combine)      (bn_stats / bn_stats_finalize)
//             pooled = pool(relu(z*inv + shift + beta)), arg-max code                          (bn_relu_pool)
//   backward: dy = pooled gradient routed to the arg-max; dbeta = sum dy; with zhat = z*inv + shift
//             dz = inv * (dy - mean(dy) - zhat * mean(dy * zhat))                               (bn_bwd_reduce / _finalize / _dz)
//             zhat at an arg-max position is (pooled - beta) wherever dy != 0, so both means come from the pooled
//             tensors alone; dz is dense and overwrites z in place, and feeds the dense-dY modes of the conv dW / dX
//             kernels.
#include "common.h"

#define BN_BLOCKS 256

// ---- forward statistics: a thread keeps one channel (strides are multiples of C)
__global__ __launch_bounds__(256) void bn_stats_kernel(const BnBatch bb) {
  __shared__ double r0[256], r1[256];
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = bb.C, active = (256 / C) * C, t = threadIdx.x;
  double s = 0.0, ss = 0.0;
  if (t < active) {
    const unsigned n = (unsigned)bb.B * bb.H * bb.W * C;         // z is dense: image stride == H*W*C
    for (unsigned e = blockIdx.x * (unsigned)active + t; e < n; e += gridDim.x * (unsigned)active) {
      const double f = (double)nb.z[e];
      s += f; ss += f * f;
    }
  }
  r0[t] = s; r1[t] = ss;
  __syncthreads();
  if (t < C) {
    double a = 0.0, b = 0.0;
    for (int k = t; k < active; k += C) { a += r0[k]; b += r1[k]; }
    nb.part[((long)blockIdx.x * 2 + 0) * C + t] = a;
    nb.part[((long)blockIdx.x * 2 + 1) * C + t] = b;
  }
}

// one 64-lane wave per (channel, network): fixed-order butterfly over the block partials
__global__ __launch_bounds__(64) void bn_stats_finalize_kernel(const BnBatch bb, int nblk, double eps) {
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = bb.C, c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { s += nb.part[((long)k * 2 + 0) * C + c]; ss += nb.part[((long)k * 2 + 1) * C + c]; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  if (threadIdx.x == 0) {
    const double count = (double)bb.B * bb.H * bb.W;
    const double mean = s / count;
    const double var = ss / count - mean * mean;     // one-pass form, as tf.nn.moments of that era
    const double inv = 1.0 / sqrt(var + eps);
    nb.stat[c] = (float)inv;
    nb.stat[C + c] = (float)(-mean * inv);
  }
}

// ---- y = z*inv + shift + beta ; relu ; 2x2 max-pool + arg-max code (first maximum wins, as everywhere else)
template <int CT>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const BnBatch bb) {
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = CT ? CT : bb.C, H = bb.H, W = bb.W, Hp = H >> 1, Wp = W >> 1;
  const unsigned ncell = (unsigned)bb.B * Hp * Wp * C;
  const long zbs = (long)H * W * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < ncell; idx += gridDim.x * 256u) {
    const int o = (int)(idx % (unsigned)C);
    unsigned r = idx / (unsigned)C;
    const int px = (int)(r % (unsigned)Wp); r /= (unsigned)Wp;
    const int py = (int)(r % (unsigned)Hp);
    const int b = (int)(r / (unsigned)Hp);
    const float inv = nb.stat[o], sh = nb.stat[C + o], be = nb.beta[o];
    const float* zp = nb.z + (long)b * zbs + ((long)(2 * py) * W + 2 * px) * C + o;
    const float v0 = (zp[0] * inv + sh) + be, v1 = (zp[C] * inv + sh) + be;
    const float v2 = (zp[(long)W * C] * inv + sh) + be, v3 = (zp[(long)W * C + C] * inv + sh) + be;
    float m = v0; int code = 0;
    if (v1 > m) { m = v1; code = 1; }
    if (v2 > m) { m = v2; code = 2; }
    if (v3 > m) { m = v3; code = 3; }
    const long e = ((long)py * Wp + px) * C + o;
    nb.pool[(long)b * nb.pool_bstride + e] = m > 0.f ? m : 0.f;
    nb.amax[(long)b * Hp * Wp * C + e] = (uint8_t)(code | (m > 0.f ? 4 : 0));      // (bit 2: POOL_ACTIVE, conv_kyo.h)
  }
}

// ---- backward reductions over the pooled tensors: part[blk][2][C] = (sum dy, sum dy * zhat) of the block's cells
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BnBatch bb) {
  __shared__ double r1[256], r2[256];
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = bb.C, active = (256 / C) * C, t = threadIdx.x;
  const int cells_per_img = (bb.H >> 1) * (bb.W >> 1) * C;
  double s1 = 0.0, s2 = 0.0;
  if (t < active) {
    const float be = nb.beta[t % C];
    const unsigned ncell = (unsigned)bb.B * cells_per_img;
    for (unsigned idx = blockIdx.x * (unsigned)active + t; idx < ncell; idx += gridDim.x * (unsigned)active) {
      const int b = (int)(idx / (unsigned)cells_per_img);
      const long e = (long)idx - (long)b * cells_per_img;
      const float pv = nb.pool[(long)b * nb.pool_bstride + e];
      if (pv > 0.f) {
        const float g = nb.dpool[(long)b * nb.dpool_bstride + e];
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
    nb.part[((long)blockIdx.x * 2 + 0) * C + t] = a1;
    nb.part[((long)blockIdx.x * 2 + 1) * C + t] = a2;
  }
}

// means[2][C] = (sum dy, sum dy*zhat) / N; dbeta = sum dy
__global__ __launch_bounds__(64) void bn_bwd_finalize_kernel(const BnBatch bb, int nblk) {
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = bb.C, c = blockIdx.x;
  double a1 = 0.0, a2 = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) { a1 += nb.part[((long)k * 2 + 0) * C + c]; a2 += nb.part[((long)k * 2 + 1) * C + c]; }
  for (int o = 32; o > 0; o >>= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
  if (threadIdx.x == 0) {
    const double n = (double)bb.B * bb.H * bb.W;
    nb.means[c] = (float)(a1 / n);
    nb.means[C + c] = (float)(a2 / n);
    nb.dbeta[c] = (float)a1;
  }
}

// z (plain conv output) -> dz in place
template <int CT>
__global__ __launch_bounds__(256) void bn_bwd_dz_kernel(const BnBatch bb) {
  const BnNet& nb = bb.n[blockIdx.y];
  const int C = CT ? CT : bb.C, H = bb.H, W = bb.W, Hp = H >> 1, Wp = W >> 1;
  const unsigned n = (unsigned)bb.B * H * W * C;
  const long zbs = (long)H * W * C;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < n; idx += gridDim.x * 256u) {
    const int o = (int)(idx % (unsigned)C);
    unsigned r = idx / (unsigned)C;
    const int x = (int)(r % (unsigned)W); r /= (unsigned)W;
    const int y = (int)(r % (unsigned)H);
    const int b = (int)(r / (unsigned)H);
    float* zp = nb.z + (long)b * zbs + ((long)y * W + x) * C + o;
    const float inv = nb.stat[o];
    const float zhat = *zp * inv + nb.stat[C + o];
    float dy = 0.f;
    const int py = y >> 1, px = x >> 1;
    if (py < Hp && px < Wp) {
      const long e = ((long)py * Wp + px) * C + o;
      if (nb.amax[(long)b * Hp * Wp * C + e] == (uint8_t)(4 | ((y & 1) * 2 + (x & 1))))      // (bit 2: the pooled output is > 0)
        dy = nb.dpool[(long)b * nb.dpool_bstride + e];
    }
    *zp = inv * ((dy - nb.means[o]) - zhat * nb.means[C + o]);
  }
}

size_t bn_part_doubles(int C) { return (size_t)BN_BLOCKS * 2 * C; }

static int ew_grid(long n, int cap) {
  long g = (n + 255) / 256;
  if (g > cap) g = cap;
  return g < 1 ? 1 : (int)g;
}

// statistics of z + BN/ReLU/pool for bb.count networks
int launch_bn_forward(cpp_ctx* ctx, const BnBatch& bb, double eps) {
  prof_begin(ctx);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(BN_BLOCKS, bb.count), dim3(256), 0, ctx->stream, bb);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(bb.C, bb.count), dim3(64), 0, ctx->stream, bb, BN_BLOCKS, eps);
  const long ncell = (long)bb.B * (bb.H / 2) * (bb.W / 2) * bb.C;
  const int grid = ew_grid(ncell, 4096);
  if (bb.C == 10) hipLaunchKernelGGL(bn_relu_pool_kernel<10>, dim3(grid, bb.count), dim3(256), 0, ctx->stream, bb);
  else hipLaunchKernelGGL(bn_relu_pool_kernel<0>, dim3(grid, bb.count), dim3(256), 0, ctx->stream, bb);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// reductions + dz (over z) for bb.count networks
int launch_bn_backward(cpp_ctx* ctx, const BnBatch& bb) {
  prof_begin(ctx);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(BN_BLOCKS, bb.count), dim3(256), 0, ctx->stream, bb);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bb.C, bb.count), dim3(64), 0, ctx->stream, bb, BN_BLOCKS);
  const int grid = ew_grid((long)bb.B * bb.H * bb.W * bb.C, 8192);
  if (bb.C == 10) hipLaunchKernelGGL(bn_bwd_dz_kernel<10>, dim3(grid, bb.count), dim3(256), 0, ctx->stream, bb);
  else hipLaunchKernelGGL(bn_bwd_dz_kernel<0>, dim3(grid, bb.count), dim3(256), 0, ctx->stream, bb);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

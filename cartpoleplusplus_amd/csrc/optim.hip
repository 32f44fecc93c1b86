// Optimiser kernels over the flat parameter buffers: global-norm clip (util.py:45-50,
// tf.clip_by_global_norm) fused with the SGD apply (ddpg_cartpole.py:118-119, :213, :218), and the
// target-network soft update (base_network.py:20-33).  Both gradient lists (actor, critic) are
// handled by one launch each (blockIdx.y = list).  Reductions are two-stage and fixed-order.
#include "common.h"

constexpr int OPT_THREADS = 256;

// part[seg][blk] = sum over the block's slice of (grad_scale * g)^2, in f64
__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const Seg2 s, float grad_scale,
                                                            double* part, int nparts) {
  __shared__ double red[OPT_THREADS];
  const int seg = blockIdx.y;
  const float* g = s.g[seg];
  const long n = s.n[seg];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)nparts * OPT_THREADS) {
    const float v = g[i] * grad_scale;
    acc += (double)v * (double)v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = OPT_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[seg * nparts + blockIdx.x] = red[0];
}

int launch_sumsq(cpp_ctx* ctx, const Seg2& s, float grad_scale, double* part, int nparts) {
  prof_begin(ctx);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nparts, 2), dim3(OPT_THREADS), 0, ctx->stream, s, grad_scale,
                     part, nparts);
  LAUNCH_CHECK();
  prof_end(ctx, K_SUMSQ);
  return 0;
}

// g <- g * clip * min(1/norm, 1/clip);  p <- p - lr * g      (clip <= 0: no clipping)
__global__ __launch_bounds__(OPT_THREADS) void clip_sgd_kernel(const Seg2 s, float grad_scale,
                                                               float clip, const double* part,
                                                               int nparts, float* norms_out) {
  __shared__ float sh_scale;
  const int seg = blockIdx.y;
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (int i = 0; i < nparts; ++i) tot += part[seg * nparts + i];
    const float norm = (float)sqrt(tot);
    float sc = 1.f;
    if (clip > 0.f) sc = clip * fminf(1.f / norm, 1.f / clip);
    sh_scale = sc * grad_scale;
    if (blockIdx.x == 0 && norms_out) norms_out[seg] = norm;
  }
  __syncthreads();
  const float sc = sh_scale, lr = s.lr[seg];
  float* p = s.p[seg];
  const float* g = s.g[seg];
  const long n = s.n[seg];
  for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * OPT_THREADS)
    p[i] = p[i] - lr * (g[i] * sc);
}

int launch_clip_sgd(cpp_ctx* ctx, const Seg2& s, float grad_scale, float clip, const double* part,
                    int nparts, float* norms_out) {
  prof_begin(ctx);
  hipLaunchKernelGGL(clip_sgd_kernel, dim3(128, 2), dim3(OPT_THREADS), 0, ctx->stream, s, grad_scale,
                     clip, part, nparts, norms_out);
  LAUNCH_CHECK();
  prof_end(ctx, K_CLIP_SGD);
  return 0;
}

// target.assign_sub(coeff * (target - source)) for every variable of the namespace
__global__ void soft_update_kernel(float* t0, const float* s0, long n0, float* t1, const float* s1,
                                   long n1, float coeff) {
  float* t = blockIdx.y == 0 ? t0 : t1;
  const float* s = blockIdx.y == 0 ? s0 : s1;
  const long n = blockIdx.y == 0 ? n0 : n1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float tv = t[i];
    t[i] = tv - coeff * (tv - s[i]);
  }
}

int launch_soft_update(cpp_ctx* ctx, float* t0, const float* s0, long n0, float* t1, const float* s1,
                       long n1, float coeff) {
  prof_begin(ctx);
  hipLaunchKernelGGL(soft_update_kernel, dim3(128, t1 ? 2 : 1), dim3(256), 0, ctx->stream, t0, s0, n0,
                     t1, s1, n1, coeff);
  LAUNCH_CHECK();
  prof_end(ctx, K_SOFT_UPDATE);
  return 0;
}

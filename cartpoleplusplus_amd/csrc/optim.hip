// Optimiser kernels over the flat parameter buffers: global-norm clip (util.py:45-50,
// tf.clip_by_global_norm) fused with the optimiser apply -- plain SGD for DDPG (ddpg_cartpole.py:118-119,
// :213, :218); GradientDescent / Momentum / Adam for NAF (util.py:73-76, naf_cartpole.py:233-239) -- and the
// target-network soft update (base_network.py:20-33).  All gradient lists are handled by one launch each
// (blockIdx.y = segment).  Reductions are two-stage and fixed-order.
#include "common.h"
#include "stats_body.h"

constexpr int OPT_THREADS = 256;

// part[seg][blk] = sum over the block's slice of (grad_scale * g)^2, in f64
__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const OptSegs s, float grad_scale,
                                                            double* part, int nparts) {
  __shared__ double red[OPT_THREADS];
  const int seg = blockIdx.y;
  const float* g = s.g[seg];
  const long n = s.n[seg];
  double acc = 0.0;
  const long stride = (long)nparts * OPT_THREADS;
  long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {      // 8 loads in flight; same order of additions as the plain loop
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = g[i + u * stride] * grad_scale;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (double)v[u] * (double)v[u];
  }
  for (; i < n; i += stride) {
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

int launch_sumsq(cpp_ctx* ctx, const OptSegs& s, float grad_scale, double* part, int nparts) {
  prof_begin(ctx);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nparts, s.nseg), dim3(OPT_THREADS), 0, ctx->stream, s, grad_scale,
                     part, nparts);
  LAUNCH_CHECK();
  prof_end(ctx, K_SUMSQ);
  return 0;
}

// g <- g * clip * min(1/norm, 1/clip) with norm over the segment's group (clip <= 0: no clipping), then
//   SGD      : p -= lr * g
//   Momentum : m = momentum * m + g;  p -= lr * m
//   Adam     : lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m, v moments;  p -= lr_t * m / (sqrt(v) + eps)
__global__ __launch_bounds__(OPT_THREADS) void opt_apply_kernel(const OptSegs s, float grad_scale,
                                                                float clip, const double* part,
                                                                int nparts, float* norms_out) {
  __shared__ float sh_scale, sh_lr;
  const int seg = blockIdx.y;
  if (seg == s.nseg) {                               // the rider's grid row (uniform per workgroup)
    const int job = (int)blockIdx.x * (OPT_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (job < s.st_jobs) stats_finalize_wave(s.st_part, s.st_nparts, s.st_C, s.st_count, s.st_white, s.st_eps, job, (int)(threadIdx.x & 63));
    return;
  }
  if (s.skip_if && *s.skip_if) return;               // (uniform)
  // squared norm of the segment's group: the first wave adds the partials (one load per lane and segment in flight, then a
  // fixed-order butterfly) -- a one-thread loop over them was most of this kernel's time
  double tot = 0.0;
  if (threadIdx.x < 64) {
    if (s.sq) {          // partials left by the kernels that wrote the gradients (cpp_ctx::sq_part), slot order
      const int gr = s.group[seg];
      const double* q = s.sq + s.sq_begin[gr];
      for (int i = threadIdx.x; i < s.sq_count[gr]; i += 64) tot += q[i];
    } else {
      for (int k = 0; k < s.nseg; ++k)
        if (s.group[k] == s.group[seg])
          for (int i = threadIdx.x; i < nparts; i += 64) tot += part[k * nparts + i];
    }
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(tot);
    float sc = 1.f;
    if (clip > 0.f) sc = clip * fminf(1.f / norm, 1.f / clip);
    sh_scale = sc * grad_scale;
    float lr = s.lr[seg];
    if (s.kind == OPT_ADAM) {
      const double t = (double)(*s.step);
      lr = (float)((double)lr * sqrt(1.0 - pow((double)s.beta2, t)) / (1.0 - pow((double)s.beta1, t)));
    }
    sh_lr = lr;
    if (blockIdx.x == 0 && norms_out && s.n[seg] > 0) norms_out[s.group[seg]] = norm;
    if (blockIdx.x == 0 && seg == 0 && s.bump) *s.bump += 1;
  }
  __syncthreads();
  const float sc = sh_scale, lr = sh_lr;
  float* p = s.p[seg];
  const float* g = s.g[seg];
  const long n = s.n[seg];
  const long stride = (long)gridDim.x * OPT_THREADS;
  if (s.kind == OPT_SGD) {
    long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
      float pv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { pv[u] = p[i + u * stride]; gv[u] = g[i + u * stride]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) p[i + u * stride] = pv[u] - lr * (gv[u] * sc);
    }
    for (; i < n; i += stride)
      p[i] = p[i] - lr * (g[i] * sc);
  } else if (s.kind == OPT_MOMENTUM) {
    float* m = s.m[seg];
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += stride) {
      const float acc = s.momentum * m[i] + g[i] * sc;
      m[i] = acc;
      p[i] = p[i] - lr * acc;
    }
  } else {
    float* m = s.m[seg];
    float* v = s.v[seg];
    const float b1 = s.beta1, b2 = s.beta2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += stride) {
      const float gi = g[i] * sc;
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      p[i] = p[i] - lr * mi / (sqrtf(vi) + s.epsilon);
    }
  }
}

int launch_opt_apply(cpp_ctx* ctx, const OptSegs& s, float grad_scale, float clip, const double* part,
                     int nparts, float* norms_out) {
  prof_begin(ctx);
  hipLaunchKernelGGL(opt_apply_kernel, dim3(128, s.nseg + (s.st_part ? 1 : 0)), dim3(OPT_THREADS), 0, ctx->stream, s, grad_scale,
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

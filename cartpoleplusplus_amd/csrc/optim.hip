// Optimiser kernels over the flat parameter buffers: global-norm clip (util.py:45-50,
// tf.clip_by_global_norm) fused with the optimiser apply -- plain SGD for DDPG (ddpg_cartpole.py:118-119,
// :213, :218); GradientDescent / Momentum / Adam for NAF (util.py:73-76, naf_cartpole.py:233-239) -- and the
// target-network soft update (base_network.py:20-33).  All gradient lists are handled by one launch each
// (blockIdx.y = segment).  Reductions are two-stage and fixed-order.
#include "common.h"
#include "stats_body.h"
#include "conv_rs16.h"

constexpr int OPT_THREADS = 256;
static_assert(OPT_THREADS == CONV_THREADS, "conv1_image_body (the rider in opt_apply_kernel) strides its loops by CONV_THREADS and assumes four waves");

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

// a lane's share of a list of f64 partials, added in list order -- eight loads in flight at a time (a `tot += q[i]` loop over ~500
// partials was eight dependent round trips in front of every update: the launch's longest chain)
__device__ __forceinline__ double lane_sum_f64(const double* q, const int count, const int lane, double tot = 0.0) {
  for (int i = lane; i < count; i += 8 * 64) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = i + 64 * u; v[u] = q[j < count ? j : i]; }      // (no branch around a load)
#pragma unroll
    for (int u = 0; u < 8; ++u) tot += (i + 64 * u < count) ? v[u] : 0.0;      // (x + 0.0 is x: squares are never -0.0)
  }
  return tot;
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
  if (s.img_n > 0 && seg == s.nseg + (s.st_part ? 1 : 0)) {      // conv1's operand images of the next minibatch (uniform per workgroup)
    if ((int)blockIdx.x >= s.img_n) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char img_lds[];
    __shared__ float wh[2 * CPP_MAX_CHANNELS];
    const int j = blockIdx.x, ws = s.img[j].seg;
    // (run by conv1_image_body once its weight / gradient loads are in flight)
    auto pre = [&](float) __attribute__((always_inline)) -> float {
    // the update's scale, as the workgroups of segment ws compute it below (same partials, same order)
      if (s.img[j].gw) {
        double tot = 0.0;
        if (threadIdx.x < 64) {
          if (s.sq) {
            const int gr = s.group[ws];
            const double* q = s.sq + s.sq_begin[gr];
            tot = lane_sum_f64(q, s.sq_count[gr], (int)threadIdx.x);
          } else {
            for (int k = 0; k < s.nseg; ++k)
              if (s.group[k] == s.group[ws])
                tot = lane_sum_f64(part + k * nparts, nparts, (int)threadIdx.x, tot);
          }
          for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        }
        if (threadIdx.x == 0) {
          const float norm = (float)sqrt(tot);
          float sc = 1.f;
          if (clip > 0.f) sc = clip * fminf(1.f / norm, 1.f / clip);
          sh_scale = sc * grad_scale;
        }
      }
      // the whitening table of the network's state column, as the first rider's waves compute it (stats_finalize_wave: a lane's rows in
      // increasing order, then the butterfly) -- with every load of the wave's channels in flight at once (channel after channel the
      // five round trips were 10 us of this workgroup)
      if (s.img[j].white) {
        if ((int)threadIdx.x < 2 * s.img_cin) wh[threadIdx.x] = s.img[j].white[threadIdx.x];      // (finished by the dW reductions' launch)
      } else {
        constexpr int NW = OPT_THREADS / 64, MAXC = (18 + NW - 1) / NW, MAXR = 8;
        const int wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63), C = s.st_C, np_ = s.st_nparts;
        if (np_ <= 64 * MAXR) {
          double v[MAXC][MAXR][2];
#pragma unroll
          for (int k = 0; k < MAXC; ++k) {
            const int c = wv + NW * k;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) {
              const int b = ln + 64 * r;
              const bool ok = c < C && b < np_;
              const double* q = s.st_part + ((long)s.img[j].col * np_ + (ok ? b : 0)) * 2 * C;
              v[k][r][0] = ok ? q[c] : 0.0; v[k][r][1] = ok ? q[C + c] : 0.0;
            }
          }
#pragma unroll
          for (int k = 0; k < MAXC; ++k) {
            const int c = wv + NW * k;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int r = 0; r < MAXR; ++r) if (ln + 64 * r < np_) { a0 += v[k][r][0]; a1 += v[k][r][1]; }
            for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
            if (ln == 0 && c < C) white_from_moments(a0, a1, s.st_count, s.st_eps, &wh[c], &wh[C + c]);
          }
        } else {
          for (int c = wv; c < C; c += NW)
            stats_finalize_wave_to(s.st_part, np_, C, s.st_count, &wh[c], &wh[C + c], s.st_eps, s.img[j].col, c, ln);
        }
      }
      __syncthreads();
      return s.img[j].gw ? sh_scale : 0.f;
    };
    Conv1ImageArgs ia;
    ia.w = s.img[j].w; ia.bias = s.img[j].bias; ia.scale = wh; ia.shift = wh + s.img_cin; ia.wscale = 0.f; ia.nout = s.img[j].nout; ia.rec = s.img[j].rec;
    ia.gw = s.img[j].gw; ia.gb = s.img[j].gb; ia.lr = s.img[j].gw ? s.lr[ws] : 0.f; ia.gscale = 0.f;      // (the scale: pre's return value)
    ia.w_out = s.img[j].gw ? s.img[j].w : nullptr; ia.b_out = s.img[j].gw ? s.img[j].bias : nullptr;
    ia.mw = (s.img[j].gw && s.kind == OPT_MOMENTUM) ? s.img[j].mw : nullptr; ia.mb = (s.img[j].gw && s.kind == OPT_MOMENTUM) ? s.img[j].mb : nullptr; ia.momentum = s.momentum;
    switch (s.img_cin) {                                // (uniform: one of conv_fwd_rs16.hip's instances)
      case 3: conv1_image_body<3, F16_PIECES>(ia, img_lds, pre); break;
      case 6: conv1_image_body<6, F16_PIECES>(ia, img_lds, pre); break;
      case 9: conv1_image_body<9, F16_PIECES>(ia, img_lds, pre); break;
      case 12: conv1_image_body<12, F16_PIECES>(ia, img_lds, pre); break;
      default: conv1_image_body<18, F16_PIECES>(ia, img_lds, pre); break;
    }
    return;
  }
  if (s.st_part && seg == s.nseg) {                  // the rider's grid row (uniform per workgroup)
    const int job = (int)blockIdx.x * (OPT_THREADS / 64) + (int)(threadIdx.x >> 6);
    if (job < s.st_jobs) stats_finalize_wave(s.st_part, s.st_nparts, s.st_C, s.st_count, s.st_white, s.st_eps, job, (int)(threadIdx.x & 63), s.st_wmax);
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
      tot = lane_sum_f64(q, s.sq_count[gr], (int)threadIdx.x);
    } else {
      for (int k = 0; k < s.nseg; ++k)
        if (s.group[k] == s.group[seg])
          tot = lane_sum_f64(part + k * nparts, nparts, (int)threadIdx.x, tot);
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
    if (blockIdx.x == 0 && seg == 0 && s.pub_wmax) route_publish_device(s.pub_wmax, s.pub_tag, s.pub_pin);
  }
  __syncthreads();
  const float sc = sh_scale, lr = sh_lr;
  // (with the image rider the segment's leading conv1 parameters are updated by its image workgroup, which needs them before and after)
  const long skip = s.img_n > 0 ? s.img_skip[seg] : 0;
  float* p = s.p[seg] + skip;
  const float* g = s.g[seg] + skip;
  const long n = s.n[seg] - skip;
  const long stride = (long)gridDim.x * OPT_THREADS;
  float* tp = s.tgt[seg] ? s.tgt[seg] + skip : nullptr;      // (uniform) the segment's target network: updated from the new values
  const float tc = s.tgt_coeff;
  if (s.kind == OPT_SGD) {
    long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
      float pv[4], gv[4], tv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { pv[u] = p[i + u * stride]; gv[u] = g[i + u * stride]; if (tp) tv[u] = tp[i + u * stride]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float pn = sgd_update(pv[u], gv[u], sc, lr);
        p[i + u * stride] = pn;
        if (tp) tp[i + u * stride] = soft_update_value(tv[u], pn, tc);
      }
    }
    for (; i < n; i += stride) {
      const float pn = sgd_update(p[i], g[i], sc, lr);
      p[i] = pn;
      if (tp) tp[i] = soft_update_value(tp[i], pn, tc);
    }
  } else if (s.kind == OPT_MOMENTUM) {
    float* m = s.m[seg] + skip;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += stride) {
      const float acc = momentum_accum(m[i], g[i], sc, s.momentum);
      m[i] = acc;
      const float pn = momentum_step(p[i], acc, lr);
      p[i] = pn;
      if (tp) tp[i] = soft_update_value(tp[i], pn, tc);
    }
  } else {
    float* m = s.m[seg] + skip;
    float* v = s.v[seg] + skip;
    const float b1 = s.beta1, b2 = s.beta2;
    for (long i = (long)blockIdx.x * OPT_THREADS + threadIdx.x; i < n; i += stride) {
      const float gi = g[i] * sc;
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      const float pn = p[i] - lr * mi / (sqrtf(vi) + s.epsilon);
      p[i] = pn;
      if (tp) tp[i] = soft_update_value(tp[i], pn, tc);
    }
  }
}

static_assert(Rs16ImageLds<18>::BYTES >= Rs16ImageLds<12>::BYTES && Rs16ImageLds<18>::BYTES >= Rs16ImageLds<9>::BYTES &&
              Rs16ImageLds<18>::BYTES >= Rs16ImageLds<6>::BYTES && Rs16ImageLds<18>::BYTES >= Rs16ImageLds<3>::BYTES, "the rider's LDS is sized for the largest instance");
int launch_opt_apply(cpp_ctx* ctx, const OptSegs& s, float grad_scale, float clip, const double* part,
                     int nparts, float* norms_out) {
  prof_begin(ctx);
  const bool img = s.img_n > 0 && (s.st_part || s.img[0].white);
  const size_t lds = img ? (size_t)Rs16ImageLds<18>::BYTES : 0;
  if (img) {
    if ((s.kind != OPT_SGD && s.kind != OPT_MOMENTUM) || (s.st_part && s.st_C != s.img_cin) || !conv_rs16_channels_ok(s.img_cin) || s.skip_if) {
      cpp_set_error("opt_apply: the conv1 image rider needs SGD or Momentum and a channel count conv_rs16.h is instantiated for (%d)", s.img_cin); return 1;
    }
    static bool attr_done[CPP_MAX_DEVICES] = {};
    if (!attr_done[cpp_dev_slot(ctx)]) {
      HIP_CHECK(hipFuncSetAttribute((const void*)opt_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Rs16ImageLds<18>::BYTES));
      attr_done[cpp_dev_slot(ctx)] = true;
    }
  }
  hipLaunchKernelGGL(opt_apply_kernel, dim3(128, s.nseg + (s.st_part ? 1 : 0) + (img ? 1 : 0)), dim3(OPT_THREADS), lds, ctx->stream, s, grad_scale,
                     clip, part, nparts, norms_out);
  LAUNCH_CHECK();
  prof_end(ctx, K_CLIP_SGD);
  return 0;
}

// target.assign_sub(coeff * (target - source)) for every variable of the namespace
__global__ void soft_update_kernel(float* t0, const float* s0, long n0, float* t1, const float* s1,
                                   long n1, float coeff, unsigned* wmax_dev, const unsigned* tag_dev, unsigned long long* pin) {
  // (rider of a training call's last launch: the largest whitening scale the call saw goes to its pinned slot, tagged with the call's
  // number, and the device word starts over -- common.h: route_publish_device; rt_core.cpp: ctx_route_update)
  if (wmax_dev && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) route_publish_device(wmax_dev, tag_dev, pin);
  float* t = blockIdx.y == 0 ? t0 : t1;
  const float* s = blockIdx.y == 0 ? s0 : s1;
  const long n = blockIdx.y == 0 ? n0 : n1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) t[i] = soft_update_value(t[i], s[i], coeff);
}

int launch_soft_update(cpp_ctx* ctx, float* t0, const float* s0, long n0, float* t1, const float* s1,
                       long n1, float coeff) {
  prof_begin(ctx);
  const bool pub = ctx->route_rider && ctx->route_pin_dev != nullptr;
  ctx->route_rider = false;
  hipLaunchKernelGGL(soft_update_kernel, dim3(128, t1 ? 2 : 1), dim3(256), 0, ctx->stream, t0, s0, n0,
                     t1, s1, n1, coeff, pub ? ctx->white_max_dev : nullptr, ctx->route_tag_dev, ctx->route_pin_dev);
  LAUNCH_CHECK();
  prof_end(ctx, K_SOFT_UPDATE);
  return 0;
}

// The whitening tables of a minibatch from the per-row sufficient statistics the gather left (base_network.py:95-99: tf.nn.moments over
// (batch, height, width), then x * inv + (-mean * inv), inv = rsqrt(var + eps)).  One 64-lane wave per (state column, channel):
// lanes stride over the per-row partials, fixed-order butterfly combine (deterministic).  Shared by stats_finalize_kernel
// (replay.hip) and the optimiser launch that carries the same work for the NEXT minibatch (optim.hip).
#pragma once
#include "common.h"

// (scale, shift) of one channel from its sums over `count` pixels: y = x * scale + shift.
//
// A channel that is CONSTANT over the minibatch (a camera that sees one colour: bullet_cartpole.py:227-236 renders whatever is in
// front of it) has variance 0, scale rsqrt(1e-6) = 1000 and a whitened value of exactly 0 at every pixel -- as the difference of
// two numbers near 1000 mu.  Kernels that fold the affine map into their weights (conv_k16.h, conv_dw16.h) would carry those two
// numbers through f32 accumulators and keep their rounding noise (measured, round 4: 4e-4 on conv1 outputs, 3e-3 relative on conv1
// weight gradients; profiles/experiments/r04_render_probe.txt).  The table of such a channel is therefore (0, 0): x * 0 + 0 is the
// exact value of (x - mu) * 1000 on this minibatch whichever way a kernel evaluates it, and its weight gradient is exactly 0, as
// in exact arithmetic.  "Constant" = the variance is zero to the rounding of its own f64 evaluation (the sums themselves are exact
// for 8-bit pixels, gather_body.h; the smallest variance a non-constant 8-bit channel can have, one pixel off by one code in
// 8.4 M, is 1.8e-12 -- about two orders above the threshold of 1.4e-14 m2 at m2 <= 1).
__device__ __forceinline__ void white_from_moments(double s, double ss, double count, double eps, float* scale, float* shift) {
  const double mean = s / count;
  const double m2 = ss / count;
  const double var = m2 - mean * mean;               // one-pass form of tf.nn.moments (r0.9-r0.11)
  if (var <= 64.0 * 2.220446049250313e-16 * m2) { *scale = 0.f; *shift = 0.f; return; }
  const double inv = 1.0 / sqrt(var + eps);
  *scale = (float)inv;
  *shift = (float)(-mean * inv);
}

// one wave: channel c of state column w, from the sampled rows' partial sums, into (*scale, *shift)
// (wmax, optional: a device word that keeps the largest whitening scale seen -- as float bits, scales are >= 0 -- for the host's choice
// of conv1 kernels: cpp_ctx::conv1_f32, rt_core.cpp)
__device__ __forceinline__ void stats_finalize_wave_to(const double* part, int nparts, int C, double count, float* scale, float* shift, double eps,
                                                       int w, int c, int lane, unsigned* wmax = nullptr) {
  double s = 0.0, ss = 0.0;
#pragma unroll 4
  for (int b = lane; b < nparts; b += 64) {
    const double* p = part + ((long)w * nparts + b) * 2 * C;
    s += p[c]; ss += p[C + c];
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  if (lane == 0) {
    white_from_moments(s, ss, count, eps, scale, shift);
    if (wmax) atomicMax(wmax, __float_as_uint(*scale));
  }
}
__device__ __forceinline__ void stats_finalize_wave(const double* part, int nparts, int C, double count, float* white, double eps,
                                                    int job, int lane, unsigned* wmax = nullptr) {
  const int w = job / C, c = job - w * C;
  stats_finalize_wave_to(part, nparts, C, count, &white[(long)w * 2 * C + c], &white[(long)w * 2 * C + C + c], eps, w, c, lane, wmax);
}

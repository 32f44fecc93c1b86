// The whitening tables of a minibatch from the per-row sufficient statistics the gather left (base_network.py:95-99: tf.nn.moments over
// (batch, height, width), then x * inv + (-mean * inv), inv = rsqrt(var + eps)).  One 64-lane wave per (state column, channel):
// lanes stride over the per-row partials, fixed-order butterfly combine (deterministic).  Shared by stats_finalize_kernel
// (replay.hip) and the optimiser launch that carries the same work for the NEXT minibatch (optim.hip).
#pragma once
#include "common.h"

__device__ __forceinline__ void stats_finalize_wave(const double* part, int nparts, int C, double count, float* white, double eps,
                                                    int job, int lane) {
  const int w = job / C, c = job - w * C;
  double s = 0.0, ss = 0.0;
#pragma unroll 4
  for (int b = lane; b < nparts; b += 64) {
    const double* p = part + ((long)w * nparts + b) * 2 * C;
    s += p[c]; ss += p[C + c];
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  if (lane == 0) {
    const double mean = s / count;
    const double var = ss / count - mean * mean;     // one-pass form of tf.nn.moments (r0.9-r0.11)
    const double inv = 1.0 / sqrt(var + eps);
    white[(long)w * 2 * C + c] = (float)inv;
    white[(long)w * 2 * C + C + c] = (float)(-mean * inv);
  }
}

// conv3 forward for 16x16 inputs with the whole zero-haloed image in LDS: constants and the compute part (conv3_img.hip runs it
// as a kernel of its own; the conv2 forward kernel of conv_k16.h runs it as its tail on the pooled rows it has just produced).
#pragma once
#include "common.h"

#define C3_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int C3_H = 16, C3_C = 10, C3_NO = 10, C3_KS = 3, C3_K = C3_KS * C3_KS * C3_C;      // 90
constexpr int C3_STEPS = (C3_K + 3) / 4;                                                      // 23
constexpr int C3_PW = C3_H + 2, C3_IMGF = C3_PW * C3_PW * C3_C;                               // padded image: 3240 floats
constexpr int C3_IPW = 2;                                                                     // images per workgroup

// B operands of the 23 k-steps (W[k = 4 st + lj][o = li]), the k-th tap's float offset from a window's corner, and the bias
struct Conv3Ops { float bw[C3_STEPS]; int offk[C3_STEPS]; float bias; };
__device__ __forceinline__ void conv3_load_ops(Conv3Ops& o, const float* w, const float* bias, int nout, int li, int lj) {
#pragma unroll
  for (int st = 0; st < C3_STEPS; ++st) {
    const int k = 4 * st + lj;
    const bool ok = k < C3_K && li < nout;
    o.bw[st] = w[ok ? k * nout + li : 0];
    if (!ok) o.bw[st] = 0.f;
    const int kc = k < C3_K ? k : 0;
    const int ky = kc / (C3_KS * C3_C), r = kc - ky * (C3_KS * C3_C);
    o.offk[st] = (ky * C3_PW) * C3_C + r;                      // (kx, c) are contiguous in a padded row
  }
  o.bias = li < nout ? bias[li] : 0.f;
}

// one wave = (image im of the workgroup's pair, upper / lower half): four pairs of rows from the padded image in LDS ->
// pooled f32 + arg-max codes (same selection rules as the row kernels: the later candidate wins only when strictly greater)
__device__ __forceinline__ void conv3_img_half(const Conv3Ops& o, const float* img_im, int half, float* out, uint8_t* amax, int nout,
                                               int li, int lj) {
  const float* base = img_im + li * C3_C;                      // window corner of pixel x = li in padded row 0
  constexpr int Hp = C3_H / 2;
#pragma unroll 1
  for (int pp = 0; pp < 4; ++pp) {
    const int py = half * 4 + pp, y = 2 * py;
    const float* r0 = base + y * C3_PW * C3_C;                 // output row y reads padded rows y .. y + 2
    f32x4 acc0 = {o.bias, o.bias, o.bias, o.bias}, acc1 = {o.bias, o.bias, o.bias, o.bias};
#pragma unroll
    for (int st = 0; st < C3_STEPS; ++st) {
      const float a0 = r0[o.offk[st]], a1 = r0[o.offk[st] + C3_PW * C3_C];
      acc0 = C3_MFMA16(a0, o.bw[st], acc0);
      acc1 = C3_MFMA16(a1, o.bw[st], acc1);
    }
    if (li < nout) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float t0 = acc0[2 * h], t1 = acc0[2 * h + 1], u0 = acc1[2 * h], u1 = acc1[2 * h + 1];
        const float top = t1 > t0 ? t1 : t0, bot = u1 > u0 ? u1 : u0;
        const int ct = t1 > t0 ? 1 : 0, cb = u1 > u0 ? 1 : 0;
        const bool lower = bot > top;
        const float mx = lower ? bot : top;
        const int code = lower ? 2 + cb : ct;
        const int px = 2 * lj + h;
        out[(py * Hp + px) * nout + li] = mx > 0.f ? mx : 0.f;
        amax[(py * Hp + px) * nout + li] = (uint8_t)(code | (mx > 0.f ? POOL_ACTIVE : 0));
      }
    }
  }
}

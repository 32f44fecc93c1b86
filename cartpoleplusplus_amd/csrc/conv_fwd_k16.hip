// conv1 forward on the f16 matrix pipes with f32-exact operands (conv_k16.h): instantiations + geometry selection.
#include <cstdlib>
#include "conv_k16.h"

// Two bands of output rows per image when whole images would leave half of the workgroup slots empty (NAF's two trunks, a single
// network's forward): the first band's height r0 is even (pool pairs) with r0 - 2 a multiple of 5 (the band's row walk starts two
// input rows early and must start on the unrolled loop's period), as close to H / 2 as that allows.  0: no such height.
static int k16_band_rows(int H) {
  int best = 0;
  for (int r0 = 2; r0 < H; r0 += 2)
    if ((r0 - 2) % 5 == 0 && (best == 0 || abs(2 * r0 - H) < abs(2 * best - H))) best = r0;
  return (best >= 8 && H - best >= 8) ? best : 0;
}
static ConvArgsN k16_with_bands(cpp_ctx* ctx, const ConvArgsN& a, int ipw) {
  static const bool no_bands = cpp_switch_off("CPP_CONV_BANDS");
  ConvArgsN b = a;
  for (int i = 0; i < b.n; ++i) { b.a[i].nbands = 0; b.a[i].band_rows = 0; }
  if (!ctx || no_bands || a.a[0].H < 32) return b;
  const int wgs = a.n * ((a.a[0].B + ipw - 1) / ipw);          // (two workgroups are resident per CU)
  for (int i = 0; i < a.n; ++i) if (a.a[i].n3_w) return b;          // conv3 rides in this launch: whole images per workgroup
  const int r0 = k16_band_rows(a.a[0].H);
  if (wgs > ctx->num_cus || r0 == 0) return b;
  for (int i = 0; i < b.n; ++i) { b.a[i].nbands = 2; b.a[i].band_rows = r0; }
  return b;
}

#define K16_CASE(CIN_, XT_, IPW_, PLAIN_)                                                                    \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_ && plain == PLAIN_) { *handled = true; if (!ctx) return 0;      \
    if (f16_exact(ctx)) return conv_fwd_k16_launch_t<CIN_, 5, XT_, IPW_, PLAIN_, 0, F16_PIECES_EXACT>(ctx, k16_with_bands(ctx, a, IPW_)); \
    return conv_fwd_k16_launch_t<CIN_, 5, XT_, IPW_, PLAIN_>(ctx, k16_with_bands(ctx, a, IPW_)); }

int conv_fwd_k16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool plain, const ConvArgsN& a, bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (ks != 5 || in_mode != IN_F16_WHITEN || W > 128 || (W & 1) || (a.a[0].nout & 1)) return 0;
  for (int i = 0; i < a.n; ++i) {
    if (a.a[i].white_bstride != 0) return 0;          // per-image statistics would need per-image weights (f32 kernel)
    if (((uintptr_t)a.a[i].in & 3) || (a.a[i].in_bstride & 1)) return 0;      // 4-byte aligned operand loads
  }
  const int xt = W > 16 ? 2 : 1;
  const int ipw = W > 64 ? 1 : (W > 32 ? 2 : 4);
  K16_CASE(18, 2, 2, false) K16_CASE(18, 2, 2, true) K16_CASE(18, 1, 4, false) K16_CASE(18, 2, 4, false) K16_CASE(18, 2, 1, false)
  K16_CASE(6, 1, 4, false) K16_CASE(6, 2, 4, false) K16_CASE(6, 2, 2, false)
  K16_CASE(12, 2, 2, false) K16_CASE(30, 2, 1, false)
  K16_CASE(9, 2, 2, false) K16_CASE(9, 1, 4, false) K16_CASE(9, 2, 4, false) K16_CASE(9, 2, 1, false) K16_CASE(3, 2, 2, false) K16_CASE(3, 1, 4, false)
  return 0;
}

#define KB16_NINE(XT_, IPW_) if (b16_order(ctx) == B16_NINE) return conv_fwd_k16_launch_t<10, 5, XT_, IPW_, false, B16_NINE>(ctx, k16_with_bands(ctx, a, IPW_));
#define KB16_CASE(XT_, IPW_)                                                                                 \
  if (xt == XT_ && ipw == IPW_) { *handled = true; if (!ctx) return 0;                                       \
    KB16_NINE(XT_, IPW_)                                                                                       \
    return conv_fwd_k16_launch_t<10, 5, XT_, IPW_, false, B16_SIX>(ctx, k16_with_bands(ctx, a, IPW_)); }

int conv_fwd_kb16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (cin != 10 || ks != 5 || in_mode != IN_F32_PLAIN || W > 64 || (W & 1) || a.a[0].nout != KYO_NO) return 0;      // (nout even)
  for (int i = 0; i < a.n; ++i)
    if (ctx && (a.a[i].in_b16 == nullptr || a.a[i].plane_stride == 0 || (a.a[i].in_bstride & 1))) return 0;
  const int xt = W > 32 ? 2 : 1;
  const int ipw = W > 32 ? 2 : (W > 16 ? 2 : 4);
  KB16_CASE(1, 2) KB16_CASE(1, 4) KB16_CASE(2, 2)
  return 0;
}

// conv2 forward of 32x32x10 inputs (two images x two 16-pixel strips per workgroup) can run conv3 + pool3 as its tail
bool conv23_fuse_ok(int H2, int W2, int B, int nout) {
  static const bool off = cpp_switch_off("CPP_CONV23_FUSE") || cpp_switch_off("CPP_CONV3_IMG") || cpp_switch_off("CPP_CONV_B16") ||
                          cpp_switch_off("CPP_CONV_K16") || cpp_switch_off("CPP_CONV_KYO");
  return !off && H2 == 2 * C3_H && W2 == 2 * C3_H && (B % 2) == 0 && nout == KYO_NO;
}

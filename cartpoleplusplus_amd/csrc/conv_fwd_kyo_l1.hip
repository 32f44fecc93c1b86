// conv1 forward, (ky,o)-column formulation: f16/f32 image batch -> whiten -> 5x5 conv -> bias+ReLU+2x2 pool.
#include "conv_kyo.h"

#define KYO1_CASE(CIN_, XT_, IPW_)                                                                                   \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_ && chb == 16 && in_mode == IN_F16_WHITEN) { *handled = true;           \
    return conv_fwd_kyo_launch_t<CIN_, 5, XT_, IPW_, IN_F16_WHITEN>(ctx, a); }                                       \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_ && chb == 16 && in_mode == IN_F32_WHITEN) { *handled = true;           \
    return conv_fwd_kyo_launch_t<CIN_, 5, XT_, IPW_, IN_F32_WHITEN>(ctx, a); }
// rows that are only 8- or 4-byte multiples (the reference's default 50 x 50 render, odd test shapes)
#define KYO1_CASE_CHB(CIN_, XT_, IPW_, CHB_)                                                                         \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_ && chb == CHB_ && in_mode == IN_F16_WHITEN) { *handled = true;         \
    return conv_fwd_kyo_launch_t<CIN_, 5, XT_, IPW_, IN_F16_WHITEN, CHB_>(ctx, a); }                                 \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_ && chb == CHB_ && in_mode == IN_F32_WHITEN) { *handled = true;         \
    return conv_fwd_kyo_launch_t<CIN_, 5, XT_, IPW_, IN_F32_WHITEN, CHB_>(ctx, a); }

int conv_fwd_kyo_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (ks != 5 || W > 128) return 0;
  // (two column tiles per wave at W <= 64, i.e. <18, 2, 2>, halves the B-operand LDS traffic but leaves 2 waves per
  // SIMD: measured 0.391 vs 0.381 ms per step for conv1)
  const int xt = W > 64 ? 2 : 1;
  const int ipw = W > 32 ? 1 : (W > 16 ? 2 : 4);
  KYO1_CASE(18, 1, 1) KYO1_CASE(18, 1, 2) KYO1_CASE(9, 1, 1) KYO1_CASE(6, 1, 1) KYO1_CASE(6, 1, 2) KYO1_CASE(6, 1, 4)
  KYO1_CASE(30, 2, 1) KYO1_CASE(18, 2, 1)
  KYO1_CASE_CHB(18, 1, 1, 8) KYO1_CASE_CHB(6, 1, 1, 8) KYO1_CASE_CHB(9, 1, 1, 4) KYO1_CASE_CHB(9, 1, 4, 4) KYO1_CASE_CHB(12, 1, 1, 8)
  return 0;
}

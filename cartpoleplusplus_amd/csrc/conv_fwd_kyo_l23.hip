// conv2 / conv3 forward, (ky,o)-column formulation (f32 pooled activations in).
#include "conv_kyo.h"

#define KYO23_CASE(KS_, XT_, IPW_)                                                                                   \
  if (ks == KS_ && xt == XT_ && ipw == IPW_ && in_mode == IN_F32_PLAIN && chb == 16) { *handled = true;              \
    return conv_fwd_kyo_launch_t<10, KS_, XT_, IPW_, IN_F32_PLAIN>(ctx, a); }                                        \
  if (ks == KS_ && xt == XT_ && ipw == IPW_ && in_mode == IN_F32_PLAIN && chb == 8) { *handled = true;               \
    return conv_fwd_kyo_launch_t<10, KS_, XT_, IPW_, IN_F32_PLAIN, 8>(ctx, a); }                                     \
  if (ks == KS_ && xt == XT_ && ipw == IPW_ && in_mode == IN_DY) { *handled = true;                                  \
    return conv_fwd_kyo_launch_t<10, KS_, XT_, IPW_, IN_DY>(ctx, a); }                                               \
  if (ks == KS_ && xt == XT_ && ipw == IPW_ && in_mode == IN_F32_FLIP && chb == 16) { *handled = true;                \
    return conv_fwd_kyo_launch_t<10, KS_, XT_, IPW_, IN_F32_FLIP>(ctx, a); }

int conv_fwd_kyo_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (cin != 10 || (in_mode != IN_F32_PLAIN && in_mode != IN_DY && in_mode != IN_F32_FLIP) || W > 64) return 0;
  // dX rows are written whole: the strips must tile the row exactly, and all 10 columns of a block are real outputs
  if ((in_mode == IN_DY || in_mode == IN_F32_FLIP) && ((W != 16 && W != 32 && W != 64) || a.a[0].nout != KYO_NO)) return 0;
  const int xt = 1;
  const int ipw = W > 32 ? 1 : (W > 16 ? 2 : 4);
  KYO23_CASE(5, 1, 1) KYO23_CASE(5, 1, 2) KYO23_CASE(5, 1, 4)
  KYO23_CASE(3, 1, 1) KYO23_CASE(3, 1, 2) KYO23_CASE(3, 1, 4)
  return 0;
}

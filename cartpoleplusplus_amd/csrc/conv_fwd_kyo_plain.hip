// forward conv with PLAIN output rows on the (ky,o)-column kernel: the first pass of a batch-norm layer.
#include "conv_kyo.h"

#define KYOP_CASE(CIN_, KS_, XT_, IPW_, MODE_, CHB_)                                                                 \
  if (cin == CIN_ && ks == KS_ && xt == XT_ && ipw == IPW_ && in_mode == MODE_ && chb == CHB_) { *handled = true;    \
    return conv_fwd_kyo_launch_t<CIN_, KS_, XT_, IPW_, MODE_, CHB_, true>(ctx, a); }

int conv_fwd_kyo_dispatch_plain(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (W > 64 || a.a[0].nout != KYO_NO) return 0;
  const int xt = 1;
  const int ipw = W > 32 ? 1 : (W > 16 ? 2 : 4);
  KYOP_CASE(18, 5, 1, 1, IN_F16_WHITEN, 16) KYOP_CASE(18, 5, 1, 1, IN_F32_WHITEN, 16) KYOP_CASE(18, 5, 1, 1, IN_F16_WHITEN, 8)
  KYOP_CASE(6, 5, 1, 1, IN_F16_WHITEN, 8) KYOP_CASE(12, 5, 1, 1, IN_F16_WHITEN, 8)
  KYOP_CASE(10, 5, 1, 2, IN_F32_PLAIN, 16) KYOP_CASE(10, 5, 1, 2, IN_F32_PLAIN, 8) KYOP_CASE(10, 5, 1, 1, IN_F32_PLAIN, 16)
  KYOP_CASE(10, 3, 1, 4, IN_F32_PLAIN, 16) KYOP_CASE(10, 3, 1, 2, IN_F32_PLAIN, 16)
  return 0;
}

// conv1 forward, (kx,o)-column formulation (full-width tiles).
#include "conv_impl.h"

#define KXO1_CASE(CIN_)                                                                            \
  if (cin == CIN_ && in_mode == IN_F16_WHITEN) return conv_fwd_kxo_launch_t<CIN_, 5, 4, IN_F16_WHITEN>(ctx, a); \
  if (cin == CIN_ && in_mode == IN_F32_WHITEN) return conv_fwd_kxo_launch_t<CIN_, 5, 4, IN_F32_WHITEN>(ctx, a);

int conv_fwd_kxo_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a) {
  if (ks != 5 || xtw != 4) { cpp_set_error("conv1 forward (kxo): unsupported geometry ks=%d xtw=%d", ks, xtw); return 1; }
  KXO1_CASE(6) KXO1_CASE(9) KXO1_CASE(18)
  cpp_set_error("conv1 forward (kxo): unsupported channel count %d (built: 6, 9, 18)", cin);
  return 1;
}

// conv2 / conv3 forward, (kx,o)-column formulation (full-width tiles).
#include "conv_impl.h"

#define KXO23_CASE(KS_, XTW_)                                                                      \
  if (ks == KS_ && xtw == XTW_) return conv_fwd_kxo_launch_t<10, KS_, XTW_, IN_F32_PLAIN>(ctx, a);

int conv_fwd_kxo_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a) {
  if (cin != 10 || in_mode != IN_F32_PLAIN) { cpp_set_error("conv2/3 forward (kxo): cin=%d mode=%d", cin, in_mode); return 1; }
  KXO23_CASE(5, 1) KXO23_CASE(5, 2) KXO23_CASE(5, 4)
  KXO23_CASE(3, 1) KXO23_CASE(3, 2) KXO23_CASE(3, 4)
  cpp_set_error("conv2/3 forward (kxo): unsupported geometry ks=%d xtw=%d", ks, xtw);
  return 1;
}

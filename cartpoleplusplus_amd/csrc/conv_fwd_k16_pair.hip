// conv1 forward on the f16 matrix pipes, two networks per workgroup (conv_k16_pair.h): instantiations for the DDPG geometries whose
// four conv1 forwards are two pairs on the same images (actor + critic on state_1, the two targets on state_2).
#include "conv_k16_pair.h"

#define K16P_CASE(CIN_, XT_, IPW_)                                                                           \
  if (cin == CIN_ && xt == XT_ && ipw == IPW_) { *handled = true; return conv_fwd_k16_pair_launch_t<CIN_, 5, XT_, IPW_>(ctx, a); }

int conv_fwd_k16_pair_dispatch(cpp_ctx* ctx, int cin, int xt, int ipw, const ConvArgsN& a, bool* handled) {
  *handled = false;
  K16P_CASE(18, 2, 2)
  return 0;
}

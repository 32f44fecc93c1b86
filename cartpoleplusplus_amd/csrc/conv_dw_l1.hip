// conv1 dW/db instantiations (input is the whitened image batch; conv1 needs no dX).
#include "conv_impl.h"

#define DW1_CASE(CIN_)                                                                             \
  if (cin == CIN_ && in_mode == IN_F16_WHITEN)                                                     \
    return conv_dw_launch_t<CIN_, 5, 4, IN_F16_WHITEN>(ctx, a, grid);                              \
  if (cin == CIN_ && in_mode == IN_F32_WHITEN)                                                     \
    return conv_dw_launch_t<CIN_, 5, 4, IN_F32_WHITEN>(ctx, a, grid);

int conv_dw_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a,
                        int* grid) {
  if (ks != 5 || xtw != 4) {
    cpp_set_error("conv1 dW: unsupported geometry ks=%d xtw=%d", ks, xtw);
    return 1;
  }
  DW1_CASE(6) DW1_CASE(9) DW1_CASE(18) DW1_CASE(30) DW1_CASE(3) DW1_CASE(12) DW1_CASE(24) DW1_CASE(15)
  cpp_set_error("conv1 dW: unsupported channel count %d (built: 3, 6, 9, 12, 15, 18, 24, 30)", cin);
  return 1;
}

// conv1 dW / db on the f16 matrix pipes, two networks per workgroup (conv_dw16.h, NNET = 2): the actor's and the critic's conv1 reduce
// over the SAME state_1 pixels (ddpg_cartpole.py:333-334), so one workgroup stages every image row once and multiplies every A fragment
// with both networks' dY.  One instance: 18 channels, two 32-pixel chunks per row (conv_dw16.hip selects it where it was measured to pay).
#include "conv_dw16.h"

#define DW16_PAIR_CASE(CIN_, NCHK_)                                                                               \
  if (cin == CIN_ && nchk == NCHK_) { *handled = true; if (!ctx) return 0;                                      \
    if (f16_exact(ctx)) return conv_dw16_launch_t<CIN_, 5, NCHK_, false, true, F16_PIECES_EXACT>(ctx, a, grid);   \
    return conv_dw16_launch_t<CIN_, 5, NCHK_, false, true>(ctx, a, grid); }

int conv_dw16_pair_dispatch(cpp_ctx* ctx, int cin, int nchk, const ConvArgsN& a, int* grid, bool* handled) {
  *handled = false;
  DW16_PAIR_CASE(18, 2)
  return 0;
}

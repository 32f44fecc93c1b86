// dW of all three conv layers from DENSE dY rows (batch norm: the gradient w.r.t. the plain conv output is dense).
#include "conv_dw_kyo.h"

#define DWD_CASE(CIN_, KS_, NS_, MODE_, CHB_)                                                                        \
  if (cin == CIN_ && ks == KS_ && ns == NS_ && in_mode == MODE_ && chb == CHB_) { *handled = true;                   \
    return conv_dw_kyo_launch_t<CIN_, KS_, NS_, MODE_, CHB_, true>(ctx, a, grid); }

int conv_dw_kyo_dispatch_dense(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, int* grid,
                               bool* handled) {
  *handled = false;
  const int W = a.a[0].W;
  if (W > 128) return 0;
  const int ns = W > 64 ? 32 : (W > 32 ? 16 : 8);
  // conv1
  DWD_CASE(18, 5, 16, IN_F16_WHITEN, 16) DWD_CASE(18, 5, 16, IN_F32_WHITEN, 16) DWD_CASE(18, 5, 16, IN_F16_WHITEN, 8)
  DWD_CASE(6, 5, 16, IN_F16_WHITEN, 8) DWD_CASE(6, 5, 16, IN_F32_WHITEN, 16) DWD_CASE(12, 5, 16, IN_F16_WHITEN, 8)
  DWD_CASE(6, 5, 8, IN_F16_WHITEN, 16) DWD_CASE(6, 5, 8, IN_F32_WHITEN, 16)
  DWD_CASE(18, 5, 8, IN_F16_WHITEN, 16) DWD_CASE(18, 5, 8, IN_F32_WHITEN, 16)      // 18 channels at widths <= 32
  DWD_CASE(9, 5, 8, IN_F16_WHITEN, 4) DWD_CASE(9, 5, 8, IN_F32_WHITEN, 8) DWD_CASE(9, 5, 16, IN_F16_WHITEN, 16)
  DWD_CASE(30, 5, 32, IN_F16_WHITEN, 16)
  DWD_CASE(18, 5, 32, IN_F16_WHITEN, 16) DWD_CASE(6, 5, 32, IN_F16_WHITEN, 16)      // 18 / 6 channels at widths 65 .. 128
  // conv2 / conv3 (f32 pooled activations in)
  DWD_CASE(10, 5, 8, IN_F32_PLAIN, 16) DWD_CASE(10, 5, 8, IN_F32_PLAIN, 8) DWD_CASE(10, 5, 16, IN_F32_PLAIN, 16)
  DWD_CASE(10, 3, 8, IN_F32_PLAIN, 16) DWD_CASE(10, 3, 8, IN_F32_PLAIN, 8) DWD_CASE(10, 3, 16, IN_F32_PLAIN, 16)
  return 0;
}

// dW / db of the 5x5 layers, (ky,o)-column formulation: conv1 (whitened image rows in) and conv2 (f32 activations in).
#include "conv_dw_kyo.h"

#define DWKYO_CASE(CIN_, NS_, MODE_)                                                                                 \
  if (cin == CIN_ && ns == NS_ && in_mode == MODE_ && chb == 16) { *handled = true;                                  \
    return conv_dw_kyo_launch_t<CIN_, 5, NS_, MODE_>(ctx, a, grid); }
#define DWKYO_CASE_CHB(CIN_, NS_, MODE_, CHB_)                                                                       \
  if (cin == CIN_ && ns == NS_ && in_mode == MODE_ && chb == CHB_) { *handled = true;                                \
    return conv_dw_kyo_launch_t<CIN_, 5, NS_, MODE_, CHB_>(ctx, a, grid); }

int conv_dw_kyo_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, bool dense, const ConvArgsN& a, int* grid,
                         bool* handled) {
  *handled = false;
  if (dense) return conv_dw_kyo_dispatch_dense(ctx, cin, ks, in_mode, chb, a, grid, handled);
  const int W = a.a[0].W;
  if (ks != 5 || W > 128) return 0;
  const int ns = W > 64 ? 32 : (W > 32 ? 16 : 8);
  DWKYO_CASE(18, 16, IN_F16_WHITEN) DWKYO_CASE(18, 16, IN_F32_WHITEN)
  DWKYO_CASE(9, 16, IN_F16_WHITEN) DWKYO_CASE(9, 16, IN_F32_WHITEN)
  DWKYO_CASE(6, 16, IN_F16_WHITEN) DWKYO_CASE(6, 16, IN_F32_WHITEN) DWKYO_CASE(6, 8, IN_F16_WHITEN) DWKYO_CASE(6, 8, IN_F32_WHITEN)
  DWKYO_CASE(30, 32, IN_F16_WHITEN) DWKYO_CASE(30, 32, IN_F32_WHITEN)
  DWKYO_CASE(10, 8, IN_F32_PLAIN) DWKYO_CASE(10, 16, IN_F32_PLAIN)
  // the reference's default 50 x 50 render: f16 rows of 600 / 1200 / 1800 bytes
  DWKYO_CASE_CHB(18, 16, IN_F16_WHITEN, 8) DWKYO_CASE_CHB(12, 16, IN_F16_WHITEN, 8) DWKYO_CASE_CHB(6, 16, IN_F16_WHITEN, 8)
  return 0;
}

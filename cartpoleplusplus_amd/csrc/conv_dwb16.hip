// conv2 dW / db on the bf16 matrix pipes with f32-exact operands (conv_dwb16.h): instantiations + geometry selection.
#include "conv_dwb16.h"

#define DWB16_NINE(CIN_, NCHK_) if (b16_order(ctx) == B16_NINE) return conv_dwb16_launch_t<CIN_, 5, NCHK_, B16_NINE>(ctx, a, grid);
#define DWB16_CASE(CIN_, NCHK_)                                                                              \
  if (cin == CIN_ && nchk == NCHK_) { *handled = true; DWB16_NINE(CIN_, NCHK_) return conv_dwb16_launch_t<CIN_, 5, NCHK_>(ctx, a, grid); }

int conv_dwb16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, int* grid, bool* handled) {
  *handled = false;
  const int W = a.a[0].W, H = a.a[0].H;
  if (ks != 5 || in_mode != IN_F32_PLAIN || W > 64 || (W & 1) || (H & 1) || H < 4 || a.a[0].nout > KYO_NO) return 0;
  for (int i = 0; i < a.n; ++i)
    if (a.a[i].dy_dense != nullptr || ((uintptr_t)a.a[i].in & 3)) return 0;
  const int nchk = W > 32 ? 2 : 1;
  DWB16_CASE(10, 1) DWB16_CASE(10, 2)
  return 0;
}

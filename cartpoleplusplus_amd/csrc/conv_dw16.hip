// conv1 dW / db on the f16 matrix pipes with f32-exact operands (conv_dw16.h): instantiations + geometry selection.
#include "conv_dw16.h"
#include "conv_dw16_rs.h"

// (cpp_ctx_set_precision: three f16 pieces of dY instead of two)
#define DW16_CASE(CIN_, NCHK_)                                                                               \
  if (cin == CIN_ && nchk == NCHK_ && !dense) { *handled = true; if (!ctx) return 0;                         \
    if (f16_exact(ctx)) return conv_dw16_launch_t<CIN_, 5, NCHK_, false, false, F16_PIECES_EXACT>(ctx, a, grid); \
    return conv_dw16_launch_t<CIN_, 5, NCHK_>(ctx, a, grid); }
#define DW16_CASE_DENSE(CIN_, NCHK_)                                                                         \
  if (cin == CIN_ && nchk == NCHK_ && dense) { *handled = true; if (!ctx) return 0;                          \
    if (f16_exact(ctx)) return conv_dw16_launch_t<CIN_, 5, NCHK_, true, false, F16_PIECES_EXACT>(ctx, a, grid); \
    return conv_dw16_launch_t<CIN_, 5, NCHK_, true>(ctx, a, grid); }

int conv_dw16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool dense, const ConvArgsN& a, int* grid, bool* handled) {
  *handled = false;
  const int W = a.a[0].W, H = a.a[0].H;
  if (ks != 5 || in_mode != IN_F16_WHITEN || W > 128 || (W & 1) || (H & 1) || H < 4 || a.a[0].nout > KYO_NO) return 0;
  for (int i = 0; i < a.n; ++i) {
    if (a.a[i].white_bstride != 0 || (a.a[i].dy_dense != nullptr) != dense) return 0;
    if (((uintptr_t)a.a[i].in & 3) || (a.a[i].in_bstride & 1)) return 0;        // dword row staging
  }
  const int nchk = W > 64 ? 4 : (W > 32 ? 2 : 1);
  // the actor's and the critic's conv1 dW of one minibatch: one workgroup per image band serves both (CPP_DW16_PAIR=0: one each) --
  // where that was measured to pay: 18 channels at two chunks per row, and enough bands that the joint launch still puts two
  // workgroups on every CU (cfg3: 79 vs 83 us; the 50x50 render has 256 whole-image units: 82 vs 62 us; 9 channels: 57 vs 49 us --
  // profiles/experiments/r03_pairs.txt)
  static const bool no_pair = cpp_switch_off("CPP_DW16_PAIR");
  {  // 64-wide rows of 18 channels, actor + critic: one wave per (network, 32-pixel column) unit (conv_dw16_rs.h)
    const int rc = conv_dw16_rs_dispatch(ctx, cin, ks, in_mode, dense, a, grid, handled);
    if (*handled) return rc;
  }
  if (!no_pair && !dense && ctx && cin == 18 && nchk == 2 && conv_dw16_pairable(a)) {
    // (the pair kernel runs two workgroups per CU; its bands are chosen by the same one-round rule in conv_dw16_launch_t)
    const int capacity = ctx->num_cus * 2 / (a.n / 2);
    int band = (H + 1) & ~1;
    while (band > 8 && (band / 2) % 2 == 0 && a.a[0].B * ((H + band / 2 - 1) / (band / 2)) <= capacity) band /= 2;
    const int units = a.a[0].B * ((H + band - 1) / band);
    if (units >= 2 * ctx->num_cus) {
      const int rc = conv_dw16_pair_dispatch(ctx, cin, nchk, a, grid, handled);
      if (*handled) return rc;
    }
  }
  DW16_CASE_DENSE(18, 2) DW16_CASE_DENSE(6, 2) DW16_CASE_DENSE(12, 2)
  DW16_CASE(18, 2) DW16_CASE(18, 1) DW16_CASE(6, 2) DW16_CASE(6, 1) DW16_CASE(12, 2) DW16_CASE(30, 4) DW16_CASE(18, 4) DW16_CASE(9, 2) DW16_CASE(9, 1) DW16_CASE(3, 2) DW16_CASE(3, 1)
  return 0;
}

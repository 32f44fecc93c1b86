// conv2 / conv3 forward (f32 pooled activations in) and their dX kernels (same implicit GEMM with
// the rebuilt dY as input and flipped/transposed weights).
#include "conv_impl.h"

#define L23_CASE(KS_, XTW_)                                                                        \
  if (ks == KS_ && xtw == XTW_ && in_mode == IN_F32_PLAIN && epi == EPI_RELU_POOL)                 \
    return conv_fwd_launch_t<10, KS_, XTW_, IN_F32_PLAIN, EPI_RELU_POOL>(ctx, a);                  \
  if (ks == KS_ && xtw == XTW_ && in_mode == IN_DY && epi == EPI_PLAIN)                            \
    return conv_fwd_launch_t<10, KS_, XTW_, IN_DY, EPI_PLAIN>(ctx, a);                             \
  if (ks == KS_ && xtw == XTW_ && in_mode == IN_F32_PLAIN && epi == EPI_PLAIN)   /* batch norm: plain output; dX from dense dY (a.flip) */ \
    return conv_fwd_launch_t<10, KS_, XTW_, IN_F32_PLAIN, EPI_PLAIN>(ctx, a);

int conv_fwd_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, int epi,
                          const ConvArgsN& a) {
  if (cin != 10) {
    cpp_set_error("conv2/3: expected 10 input channels, got %d", cin);
    return 1;
  }
  L23_CASE(5, 1) L23_CASE(5, 2) L23_CASE(5, 4)
  L23_CASE(3, 1) L23_CASE(3, 2) L23_CASE(3, 4)
  cpp_set_error("conv2/3: unsupported geometry ks=%d xtw=%d mode=%d epi=%d", ks, xtw, in_mode, epi);
  return 1;
}

// conv1 forward instantiations: f16/f32 image batch -> whiten -> 5x5 conv -> bias+ReLU+2x2 pool.
#include "conv_impl.h"

#define L1_CASE(CIN_)                                                                              \
  if (cin == CIN_ && in_mode == IN_F16_WHITEN && epi == EPI_RELU_POOL)                             \
    return conv_fwd_launch_t<CIN_, 5, 4, IN_F16_WHITEN, EPI_RELU_POOL>(ctx, a);                    \
  if (cin == CIN_ && in_mode == IN_F32_WHITEN && epi == EPI_RELU_POOL)                             \
    return conv_fwd_launch_t<CIN_, 5, 4, IN_F32_WHITEN, EPI_RELU_POOL>(ctx, a);                    \
  if (cin == CIN_ && in_mode == IN_F16_WHITEN && epi == EPI_PLAIN)       /* batch norm: plain conv output */ \
    return conv_fwd_launch_t<CIN_, 5, 4, IN_F16_WHITEN, EPI_PLAIN>(ctx, a);                        \
  if (cin == CIN_ && in_mode == IN_F32_WHITEN && epi == EPI_PLAIN)                                 \
    return conv_fwd_launch_t<CIN_, 5, 4, IN_F32_WHITEN, EPI_PLAIN>(ctx, a);

int conv_fwd_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, int epi,
                         const ConvArgsN& a) {
  if (ks != 5 || xtw != 4) {
    cpp_set_error("conv1 forward: unsupported geometry ks=%d xtw=%d", ks, xtw);
    return 1;
  }
  L1_CASE(6) L1_CASE(9) L1_CASE(18) L1_CASE(30) L1_CASE(3) L1_CASE(12) L1_CASE(24) L1_CASE(15)
  cpp_set_error("conv1 forward: unsupported channel count %d (built: 3, 6, 9, 12, 15, 18, 24, 30)", cin);
  return 1;
}

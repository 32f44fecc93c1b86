// conv2 / conv3 dW/db instantiations + the second-stage partial reduction.
#include "conv_impl.h"

// 64 outputs x 4 slices of the partial list per workgroup; slices are combined in fixed order.  One launch
// serves every queued (layer, network) reduction: workgroup -> (descriptor, output block) via a prefix table.
#ifndef DWR_SLICES
#define DWR_SLICES 16
#endif
__global__ __launch_bounds__(64 * DWR_SLICES) void conv_dw_reduce_kernel(const DwReduceBatch rb) {
  __shared__ float red[DWR_SLICES][64];
  conv_dw_reduce_body<DWR_SLICES>(rb, (int)blockIdx.x, red);
}

int launch_dw_reduce_batch(cpp_ctx* ctx, const DwReduceBatch& rb) {
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3(rb.block_start[rb.n]), dim3(64 * DWR_SLICES), 0, ctx->stream, rb);
  LAUNCH_CHECK();
  return 0;
}

#define DW23_CASE(KS_, XTW_)                                                                       \
  if (ks == KS_ && xtw == XTW_ && in_mode == IN_F32_PLAIN)                                         \
    return conv_dw_launch_t<10, KS_, XTW_, IN_F32_PLAIN>(ctx, a, grid);

int conv_dw_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a,
                         int* grid) {
  if (cin != 10) {
    cpp_set_error("conv2/3 dW: expected 10 input channels, got %d", cin);
    return 1;
  }
  DW23_CASE(5, 1) DW23_CASE(5, 2) DW23_CASE(5, 4)
  DW23_CASE(3, 1) DW23_CASE(3, 2) DW23_CASE(3, 4)
  cpp_set_error("conv2/3 dW: unsupported geometry ks=%d xtw=%d", ks, xtw);
  return 1;
}

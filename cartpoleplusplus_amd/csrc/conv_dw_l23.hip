// conv2 / conv3 dW/db instantiations + the second-stage partial reduction.
#include "conv_impl.h"

__global__ void conv_dw_reduce_kernel(const float* __restrict__ partial, int nblocks, int pstride,
                                      int nw, int nout, float* __restrict__ grad_w,
                                      float* __restrict__ grad_b) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nw + nout) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[(long)b * pstride + e];
  if (e < nw) grad_w[e] = s; else grad_b[e - nw] = s;
}

int launch_dw_reduce(cpp_ctx* ctx, const float* partial, int nblocks, int pstride, int nw, int nout,
                     float* grad_w, float* grad_b) {
  const int n = nw + nout;
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, partial,
                     nblocks, pstride, nw, nout, grad_w, grad_b);
  LAUNCH_CHECK();
  return 0;
}

#define DW23_CASE(KS_, XTW_)                                                                       \
  if (ks == KS_ && xtw == XTW_ && in_mode == IN_F32_PLAIN)                                         \
    return conv_dw_launch_t<10, KS_, XTW_, IN_F32_PLAIN>(ctx, a, grid);

int conv_dw_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgs& a,
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

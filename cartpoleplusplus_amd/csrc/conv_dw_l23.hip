// conv2 / conv3 dW/db instantiations + the second-stage partial reduction.
#include <cstring>
#include "conv_impl.h"
#include "stats_body.h"

// 64 outputs x 4 slices of the partial list per workgroup; slices are combined in fixed order.  One launch
// serves every queued (layer, network) reduction: workgroup -> (descriptor, output block) via a prefix table.
#ifndef DWR_SLICES
#define DWR_SLICES 16
#endif
__global__ __launch_bounds__(64 * DWR_SLICES) void conv_dw_reduce_kernel(const DwReduceBatch rb, const StatsRide st) {
  __shared__ float red[DWR_SLICES][64];
  const int nb = rb.block_start[rb.n];
  if ((int)blockIdx.x >= nb) {                         // the rider: whitening tables of the next minibatch (uniform per workgroup)
    const int job = ((int)blockIdx.x - nb) * DWR_SLICES + (int)(threadIdx.x >> 6);
    if (job < st.jobs) stats_finalize_wave(st.part, st.nparts, st.C, st.count, st.white, st.eps, job, (int)(threadIdx.x & 63), st.wmax);
    return;
  }
  conv_dw_reduce_body<DWR_SLICES>(rb, (int)blockIdx.x, red);
}

int launch_dw_reduce_batch(cpp_ctx* ctx, const DwReduceBatch& rb, const StatsRide* st) {
  StatsRide s; memset(&s, 0, sizeof(s));
  if (st) s = *st;
  const int extra = st ? (st->jobs + DWR_SLICES - 1) / DWR_SLICES : 0;
  hipLaunchKernelGGL(conv_dw_reduce_kernel, dim3(rb.block_start[rb.n] + extra), dim3(64 * DWR_SLICES), 0, ctx->stream, rb, s);
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

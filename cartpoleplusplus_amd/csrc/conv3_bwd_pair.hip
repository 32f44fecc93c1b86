// conv3's dW and dX in ONE launch.  Both read the same pooled gradient, neither needs the other, and each is a latency-bound
// launch of ~12 us whose grid leaves most of the chip idle most of the time; a graph fork onto a second stream costs more
// than it hides on this runtime, two kernels in one grid do not: workgroups [0, dX grid) run the row kernel in
// dX mode, the rest the dW kernel -- the unchanged kernel bodies, told their place in a grid of their own.
#include <cstring>
#ifndef PAIR_ORDER_DEFAULT
#define PAIR_ORDER_DEFAULT 1
#endif
#include "conv_impl.h"
#include "conv_kyo.h"
#include "conv_dx_rs.h"
#include "conv_dw_rs.h"

__global__ __launch_bounds__(CONV_THREADS, 2) void conv3_bwd_pair_kernel(const ConvArgsN dx, int dx_gx, const ConvArgsN dw, int dw_gx, int order) {
  int i;
  if (!pair_grid_place((int)blockIdx.x, dx_gx * dx.n, dw_gx * dw.n, order, &i)) {
    conv_fwd_kyo_body<10, 3, 1, 4, IN_DY, 16, false>(dx, i % dx_gx, i / dx_gx);
  } else {
    conv_dw_body<10, 3, 1, IN_F32_PLAIN>(dw, i % dw_gx, i / dw_gx, dw_gx);
  }
}

// both halves on the bf16 pipes' row-streaming bodies (conv_dx_rs.h, conv_dw_rs.h; 16-wide rows)
template <int ORDER>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_bwd_pair_rs_kernel(const ConvArgsN dx, int dx_gx, const ConvArgsN dw, int dw_gx, int upi, int band, int order) {
  int i;
  if (!pair_grid_place((int)blockIdx.x, dx_gx * dx.n, dw_gx * dw.n, order, &i)) {
    conv_dx_rs_body<3, 1, ORDER>(dx, i % dx_gx, i / dx_gx);
  } else {
    conv_dw_rs_body<3, ORDER>(dw, upi, band, i % dw_gx, i / dw_gx);
  }
}

// one half on a row-streaming body, the other on the kernel it replaced (the two dispatchers decide separately: an ablation switch,
// an unaligned buffer or a dW grid beyond the partial buffers' capacity takes one half off its body and leaves the other)
template <int ORDER, bool DXRS>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv3_bwd_pair_mixed_kernel(const ConvArgsN dx, int dx_gx, const ConvArgsN dw, int dw_gx, int upi, int band, int order) {
  int i;
  if (!pair_grid_place((int)blockIdx.x, dx_gx * dx.n, dw_gx * dw.n, order, &i)) {
    if (DXRS) conv_dx_rs_body<3, 1, ORDER>(dx, i % dx_gx, i / dx_gx);
    else conv_fwd_kyo_body<10, 3, 1, 4, IN_DY, 16, false>(dx, i % dx_gx, i / dx_gx);
  } else {
    if (DXRS) conv_dw_body<10, 3, 1, IN_F32_PLAIN>(dw, i % dw_gx, i / dw_gx, dw_gx);
    else conv_dw_rs_body<3, ORDER>(dw, upi, band, i % dw_gx, i / dw_gx);
  }
}

int launch_conv3_bwd_pair(cpp_ctx* ctx, const ConvPairSlot& slot) {
  const bool dx_rs = slot.have_dx && slot.dx_rs, dw_rs = slot.have_dw && slot.dw_rs;
  if (slot.have_dx && slot.have_dw && dx_rs != dw_rs) {
    const bool nine = b16_order(ctx) == B16_NINE;
    auto kern = dx_rs ? (nine ? conv3_bwd_pair_mixed_kernel<B16_NINE, true> : conv3_bwd_pair_mixed_kernel<B16_SIX, true>)
                      : (nine ? conv3_bwd_pair_mixed_kernel<B16_NINE, false> : conv3_bwd_pair_mixed_kernel<B16_SIX, false>);
    const size_t lds = slot.dx_lds > slot.dw_lds ? slot.dx_lds : slot.dw_lds;
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    static const int order = cpp_switch_int("CPP_PAIR3_ORDER", 1);
    prof_begin(ctx);
    hipLaunchKernelGGL(kern, dim3(slot.dx_gx * slot.dx.n + slot.dw_gx * slot.dw.n), dim3(CONV_THREADS), lds, ctx->stream, slot.dx, slot.dx_gx, slot.dw, slot.dw_gx,
                       slot.upi, slot.band, order);
    LAUNCH_CHECK();
    prof_end(ctx, K_CONV3_BWD);
    return 0;
  }
  if (dx_rs || dw_rs) {
    if (!(dx_rs && dw_rs)) { cpp_set_error("conv3 backward pair: one half is missing beside a row-streaming body"); return 1; }
    const bool nine = b16_order(ctx) == B16_NINE;
    auto kern = nine ? conv3_bwd_pair_rs_kernel<B16_NINE> : conv3_bwd_pair_rs_kernel<B16_SIX>;
    const size_t lds = slot.dx_lds > slot.dw_lds ? slot.dx_lds : slot.dw_lds;
    static bool attr_done[CPP_MAX_DEVICES][2] = {};
    if (!attr_done[cpp_dev_slot(ctx)][nine ? 1 : 0]) {
      HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_done[cpp_dev_slot(ctx)][nine ? 1 : 0] = true;
    }
    static const int order = cpp_switch_int("CPP_PAIR3_ORDER", 1);
    prof_begin(ctx);
    hipLaunchKernelGGL(kern, dim3(slot.dx_gx * slot.dx.n + slot.dw_gx * slot.dw.n), dim3(CONV_THREADS), lds, ctx->stream, slot.dx, slot.dx_gx, slot.dw, slot.dw_gx,
                       slot.upi, slot.band, order);
    LAUNCH_CHECK();
    prof_end(ctx, K_CONV3_BWD);
    return 0;
  }
  const int ndx = slot.have_dx ? slot.dx_gx * slot.dx.n : 0, ndw = slot.have_dw ? slot.dw_gx * slot.dw.n : 0;
  if (ndx + ndw == 0) return 0;
  const size_t lds = (slot.have_dx ? slot.dx_lds : 0) > (slot.have_dw ? slot.dw_lds : 0) ? slot.dx_lds : slot.dw_lds;
  static size_t attr_dev[CPP_MAX_DEVICES] = {};      // (kernel attributes are per device)
  size_t& attr = attr_dev[cpp_dev_slot(ctx)];
  if (lds > attr) {
    HIP_CHECK(hipFuncSetAttribute((const void*)conv3_bwd_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  ConvArgsN dx = slot.dx, dw = slot.dw;
  if (!slot.have_dx) { memset(&dx, 0, sizeof(dx)); dx.n = 0; }
  if (!slot.have_dw) { memset(&dw, 0, sizeof(dw)); dw.n = 0; }
  static const int order = cpp_switch_int("CPP_PAIR_ORDER", PAIR_ORDER_DEFAULT);      // see conv2_bwd_pair.hip
  prof_begin(ctx);
  hipLaunchKernelGGL(conv3_bwd_pair_kernel, dim3(ndx + ndw), dim3(CONV_THREADS), lds, ctx->stream, dx, slot.have_dx ? slot.dx_gx : 1,
                     dw, slot.have_dw ? slot.dw_gx : 1, order);
  LAUNCH_CHECK();
  prof_end(ctx, K_CONV3_BWD);
  return 0;
}

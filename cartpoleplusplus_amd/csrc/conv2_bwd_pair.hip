// conv2's dW (bf16 kernel, conv_dwb16.h) and dX (row kernel in dX mode) in ONE launch, as for conv3 (conv3_bwd_pair.hip): both read
// the pooled gradient conv3's dX left, neither needs the other.  Workgroups [0, dX grid) run dX, the rest dW.
#include <cstring>
#ifndef PAIR_ORDER_DEFAULT
#define PAIR_ORDER_DEFAULT 1
#endif
#include "conv_kyo.h"
#include "conv_dwb16.h"
#include "conv_dx_rs.h"
#include "conv_dw_rs.h"

template <int ORDER>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv2_bwd_pair_kernel(const ConvArgsN dx, int dx_gx, const ConvArgsN dw, int dw_gx, int upi, int band, int order) {
  int i;
  if (!pair_grid_place((int)blockIdx.x, dx_gx * dx.n, dw_gx * dw.n, order, &i)) {
    conv_fwd_kyo_body<10, 5, 1, 2, IN_DY, 16, false>(dx, i % dx_gx, i / dx_gx);
  } else {
    conv_dwb16_body<10, 5, 1, ORDER>(dw, upi, band, i % dw_gx, i / dw_gx, dw_gx);
  }
}
// the dX half on the bf16 pipes (conv_dx_rs.h) instead of the f32-input row kernel; DWRS: the dW half one wave per unit (conv_dw_rs.h)
template <int ORDER, bool DWRS>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv2_bwd_pair_rs_kernel(const ConvArgsN dx, int dx_gx, const ConvArgsN dw, int dw_gx, int upi, int band, int order) {
  int i;
  if (!pair_grid_place((int)blockIdx.x, dx_gx * dx.n, dw_gx * dw.n, order, &i)) {
    conv_dx_rs_body<5, 2, ORDER>(dx, i % dx_gx, i / dx_gx);
  } else {
    if (DWRS) conv_dw_rs_body<5, ORDER>(dw, upi, band, i % dw_gx, i / dw_gx);
    else conv_dwb16_body<10, 5, 1, ORDER>(dw, upi, band, i % dw_gx, i / dw_gx, dw_gx);
  }
}

int launch_conv2_bwd_pair(cpp_ctx* ctx, const ConvPairSlot& slot) {
  const int ndx = slot.have_dx ? slot.dx_gx * slot.dx.n : 0, ndw = slot.have_dw ? slot.dw_gx * slot.dw.n : 0;
  if (ndx + ndw == 0) return 0;
  const size_t lds = (slot.have_dx ? slot.dx_lds : 0) > (slot.have_dw ? slot.dw_lds : 0) ? slot.dx_lds : slot.dw_lds;
  const bool nine = b16_order(ctx) == B16_NINE;      // (cpp_ctx_set_precision: every product of the bf16 pieces)
  const bool rs = slot.have_dx && slot.dx_rs;
  const bool wrs = slot.have_dw && slot.dw_rs;        // (conv_dw_rs.h only parks beside conv_dx_rs.h's dX: both dispatch on the same geometry)
  if (wrs && !rs) { cpp_set_error("conv2 backward pair: conv_dw_rs.h's dW without conv_dx_rs.h's dX"); return 1; }
  auto kern = rs ? (wrs ? (nine ? conv2_bwd_pair_rs_kernel<B16_NINE, true> : conv2_bwd_pair_rs_kernel<B16_SIX, true>)
                        : (nine ? conv2_bwd_pair_rs_kernel<B16_NINE, false> : conv2_bwd_pair_rs_kernel<B16_SIX, false>))
                 : (nine ? conv2_bwd_pair_kernel<B16_NINE> : conv2_bwd_pair_kernel<B16_SIX>);
  static size_t attr_dev[CPP_MAX_DEVICES][6] = {};   // (kernel attributes are per device and per kernel)
  size_t& attr = attr_dev[cpp_dev_slot(ctx)][(nine ? 1 : 0) + (rs ? 2 : 0) + (wrs ? 2 : 0)];
  if (lds > attr) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  ConvArgsN dx = slot.dx, dw = slot.dw;
  if (!slot.have_dx) { memset(&dx, 0, sizeof(dx)); dx.n = 0; }
  if (!slot.have_dw) { memset(&dw, 0, sizeof(dw)); dw.n = 0; }
  // the dW workgroups are few, long serial chains (a band of rows each): dispatched behind the dX grid they start when dX is nearly
  // done and the launch takes dX + dW; dispatched first they run beside it (CPP_PAIR_ORDER: 0 dX first, 1 dW first, 2 interleaved)
  // (with the bf16 dX body the picture turns round: its 256 workgroups are the long ones -- one per CU, 24 us -- and go first; the dW
  // workgroups fill the other slot of every CU beside them and both slots afterwards: 53 us against 58 dW-first, r05_dxrs_sweep.sh)
  static const int order_sw = cpp_switch_int("CPP_PAIR_ORDER", -1);
  const int order = order_sw >= 0 ? order_sw : (rs ? 0 : PAIR_ORDER_DEFAULT);
  prof_begin(ctx);
  hipLaunchKernelGGL(kern, dim3(ndx + ndw), dim3(CONV_THREADS), lds, ctx->stream, dx, slot.have_dx ? slot.dx_gx : 1,
                     dw, slot.have_dw ? slot.dw_gx : 1, slot.upi, slot.band, order);
  LAUNCH_CHECK();
  prof_end(ctx, K_CONV2_BWD);
  return 0;
}

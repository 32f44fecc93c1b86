// conv2's dW on the bf16 pipes, one wave per unit (conv_dw_rs.h): its own launch, or parked for conv2_bwd_pair.hip
#include <cstring>
#include "conv_dw_rs.h"

int conv_dw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& batch, int* grid_out, bool* handled) {
  *handled = false;
  static const bool off = cpp_switch_off("CPP_CONV_DWRS") || cpp_switch_off("CPP_CONV_B16");
  const ConvArgs& a = batch.a[0];
  static const bool off3 = cpp_switch_off("CPP_CONV3_DWRS");
  // (16-wide rows -- conv3 at 64x64 images -- fill half of the 32-pixel chunk and would leave conv3_bwd_pair.hip's launch for one of their own)
  static const bool w16 = cpp_switch_int("CPP_CONV3_DWRS_W16", 0) != 0;
  const bool geo5 = ks == 5 && (a.W == 32 || a.W == 64), geo3 = ks == 3 && !off3 && ((a.W == 16 && (w16 || conv3_pair_rs_ok(ctx, a.H, a.W))) || a.W == 32 || a.W == 64);
  if (off || !(geo5 || geo3) || cin != KYO_NO || in_mode != IN_F32_PLAIN || (a.H & 1) || a.H < 8 || a.nout != KYO_NO) return 0;
  for (int i = 0; i < batch.n; ++i)
    if (batch.a[i].dy_dense != nullptr || ((uintptr_t)batch.a[i].in & 7) || (batch.a[i].in_bstride & 1)) return 0;
  *handled = true;
  // units = (image, band of rows), one per WAVE: bands as short as filling every SIMD's slot once allows (a band reads 2 P more dZ
  // rows than it has input rows; it multiplies no row twice)
  const int capacity = ctx->num_cus * 4 / batch.n, ncol = (a.W + 31) / 32;
  int band = (a.H + 1) & ~1;
  while (band > 8 && (band / 2) % 2 == 0 && a.B * ncol * ((a.H + band / 2 - 1) / (band / 2)) <= capacity) band /= 2;
  const int upi = ((a.H + band - 1) / band) * ncol;           // (band, 32-pixel column) units of an image, column fastest
  const int grid = (a.B * upi + 3) / 4;
  // one partial per workgroup: the partial buffers hold num_cus * 4 of them per network (conv_dw_partial_floats); a launch that needs
  // more (B > 16 num_cus / ncol whole-image units: large batches, CPX partitions) is left to the kernels that clamp their grids
  if ((size_t)grid > (size_t)ctx->num_cus * 4) { *handled = false; return 0; }
  *grid_out = grid;
  const bool nine = b16_order(ctx) == B16_NINE;
  if (ctx->pair && ctx->pair->layer == 1 && ks == 5 && a.W == 32) {      // leaves with conv2's dX (conv2_bwd_pair.hip)
    ctx->pair->dw = batch; ctx->pair->dw_gx = grid; ctx->pair->dw_lds = DwRsGeom<5>::LDS_BYTES; ctx->pair->have_dw = true;
    ctx->pair->upi = upi; ctx->pair->band = band; ctx->pair->dw_rs = true;
    return 0;
  }
  if (ks == 3 && conv3_pair_rs_ok(ctx, a.H, a.W)) {             // leaves with conv3's dX (conv3_bwd_pair.hip)
    ctx->pair->dw = batch; ctx->pair->dw_gx = grid; ctx->pair->dw_lds = DwRsGeom<3>::LDS_BYTES; ctx->pair->have_dw = true;
    ctx->pair->upi = upi; ctx->pair->band = band; ctx->pair->dw_rs = true;
    return 0;
  }
  typedef void (*kern_t)(const ConvArgsN, int, int);
  const kern_t kern = ks == 5 ? (nine ? (kern_t)conv_dw_rs_kernel<5, B16_NINE> : (kern_t)conv_dw_rs_kernel<5, B16_SIX>)
                              : (nine ? (kern_t)conv_dw_rs_kernel<3, B16_NINE> : (kern_t)conv_dw_rs_kernel<3, B16_SIX>);
  const int lds = ks == 5 ? DwRsGeom<5>::LDS_BYTES : DwRsGeom<3>::LDS_BYTES;
  static bool attr_done[CPP_MAX_DEVICES][4] = {};
  bool& done = attr_done[cpp_dev_slot(ctx)][(nine ? 1 : 0) + (ks == 5 ? 0 : 2)];
  if (!done) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds, ctx->stream, batch, upi, band);
  LAUNCH_CHECK();
  return 0;
}

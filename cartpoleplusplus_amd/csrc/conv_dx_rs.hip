// conv2's dX on the bf16 pipes, row-streaming (conv_dx_rs.h): its own launch, or parked for conv2_bwd_pair.hip
#include <cstring>
#include "conv_dx_rs.h"

template <int KS, int TPR, int ORDER>
static int conv_dx_rs_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef DxRsGeom<KS, TPR> G;
  const ConvArgs& a = batch.a[0];
  auto kern = conv_dx_rs_kernel<KS, TPR, ORDER>;
  static bool attr_done[CPP_MAX_DEVICES] = {};
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int nb = a.nbands > 1 ? a.nbands : 1;
  hipLaunchKernelGGL(kern, dim3(((a.B + G::IPW - 1) / G::IPW) * nb, batch.n), dim3(CONV_THREADS), G::LDS_BYTES, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

bool conv3_pair_rs_ok(const cpp_ctx* ctx, int H, int W) {
  static const bool off = cpp_switch_off("CPP_CONV3_PAIR_RS") || cpp_switch_off("CPP_CONV_DXRS") || cpp_switch_off("CPP_CONV_DWRS") ||
                          cpp_switch_off("CPP_CONV3_DXRS") || cpp_switch_off("CPP_CONV3_DWRS") || cpp_switch_off("CPP_CONV_B16") || cpp_switch_off("CPP_CONV3_PAIR");
  return !off && ctx && ctx->pair && ctx->pair->layer == 2 && W == 16 && H >= 8 && !(H & 1);
}

bool conv_dx_rs_ok(const cpp_ctx* ctx, int cin, int ks, int H, int W, int nout) {
  // (every switch that keeps a dX launch off this file's kernels belongs HERE: conv1's dW takes its 2^S from the bounds these kernels leave
  // -- rt_net.cpp asks this predicate --, and a dX that ran elsewhere would leave stale ones)
  static const bool off = cpp_switch_off("CPP_CONV_DXRS") || cpp_switch_off("CPP_CONV_B16") || cpp_switch_off("CPP_CONV_KYO");
  (void)ctx;
  static const bool off3 = cpp_switch_off("CPP_CONV3_DXRS");
  // (16-wide rows -- conv3 at 64x64 images -- would leave conv3_bwd_pair.hip's launch for one of their own: CPP_CONV3_DXRS_W16=1, ablation build)
  static const bool w16 = cpp_switch_int("CPP_CONV3_DXRS_W16", 0) != 0;
  const bool geo = (ks == 5 && (W == 32 || W == 64)) || (ks == 3 && !off3 && ((W == 16 && (w16 || conv3_pair_rs_ok(ctx, H, W))) || W == 32 || W == 64));
  return !off && cin == KYO_NO && nout == KYO_NO && geo && H >= 4 && !(H & 1);
}

int conv_dx_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a_in, bool* handled) {
  *handled = false;
  const ConvArgs& a0 = a_in.a[0];
  if (in_mode != IN_DY || !conv_dx_rs_ok(ctx, cin, ks, a0.H, a0.W, a0.nout)) return 0;
  *handled = true;
  const bool nine = b16_order(ctx) == B16_NINE;
  ConvArgsN a = a_in;
  {  // fewer workgroups than CUs: two bands of rows per image (CPP_DXRS_BANDS=0: whole images)
    static const int bands_sw = cpp_switch_int("CPP_DXRS_BANDS", 1);      // (ablation build: 0 never, 2 always)
    const int ipw = a0.W == 16 ? 4 : (a0.W == 32 ? 2 : 1);
    const int wgs = a.n * ((a0.B + ipw - 1) / ipw);
    // (not beside conv2's dW: there the launch is bound by the two bodies' total work, and a second band walks up to NSET - 1 + 2 P rows
    // more per image -- cfg4 33.5 -> 35.5 us, cfg3 52.8 -> 57 us with bands, profiles/experiments/r05_dxrs_bands.sh)
    const bool paired = (ctx->pair && ctx->pair->layer == 1 && a0.W == 32) || (ks == 3 && conv3_pair_rs_ok(ctx, a0.H, a0.W));
    const bool two = bands_sw != 0 && ((wgs < ctx->num_cus && !paired) || bands_sw == 2) && a0.H >= 16 && (a0.H % 4) == 0;
    for (int i = 0; i < a.n; ++i) { a.a[i].nbands = two ? 2 : 1; a.a[i].band_rows = two ? a0.H / 2 : a0.H; }
  }
  if (ctx->pair && ctx->pair->layer == 1 && ks == 5 && a0.W == 32) {     // leaves with conv2's dW (conv2_bwd_pair.hip)
    ctx->pair->dx = a; ctx->pair->dx_gx = ((a0.B + 1) / 2) * a.a[0].nbands; ctx->pair->dx_lds = DxRsGeom<5, 2>::LDS_BYTES; ctx->pair->have_dx = true;
    ctx->pair->dx_rs = true;
    return 0;
  }
  if (ks == 3 && conv3_pair_rs_ok(ctx, a0.H, a0.W) && a0.nout == KYO_NO) {      // leaves with conv3's dW (conv3_bwd_pair.hip)
    ctx->pair->dx = a; ctx->pair->dx_gx = ((a0.B + 3) / 4) * a.a[0].nbands; ctx->pair->dx_lds = DxRsGeom<3, 1>::LDS_BYTES; ctx->pair->have_dx = true;
    ctx->pair->dx_rs = true;
    return 0;
  }
  if (ks == 3) {
    if (a0.W == 16) return nine ? conv_dx_rs_launch_t<3, 1, B16_NINE>(ctx, a) : conv_dx_rs_launch_t<3, 1, B16_SIX>(ctx, a);
    if (a0.W == 32) return nine ? conv_dx_rs_launch_t<3, 2, B16_NINE>(ctx, a) : conv_dx_rs_launch_t<3, 2, B16_SIX>(ctx, a);
    return nine ? conv_dx_rs_launch_t<3, 4, B16_NINE>(ctx, a) : conv_dx_rs_launch_t<3, 4, B16_SIX>(ctx, a);
  }
  if (a0.W == 32) return nine ? conv_dx_rs_launch_t<5, 2, B16_NINE>(ctx, a) : conv_dx_rs_launch_t<5, 2, B16_SIX>(ctx, a);
  return nine ? conv_dx_rs_launch_t<5, 4, B16_NINE>(ctx, a) : conv_dx_rs_launch_t<5, 4, B16_SIX>(ctx, a);
}

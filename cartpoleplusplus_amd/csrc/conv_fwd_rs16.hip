// conv1 forward, row-streaming with the weights in registers (conv_rs16.h): instantiations + geometry selection.
#include <cstdlib>
#include <cstring>
#include "conv_rs16.h"

template <int CIN>
static int conv_fwd_rs16_launch(cpp_ctx* ctx, const ConvArgsN& a, const Conv1ImageArgsN* ia) {
  if (ia) {
    constexpr int ilds = Rs16ImageLds<CIN>::BYTES;
    static bool attr_done[CPP_MAX_DEVICES] = {};
    if (!attr_done[cpp_dev_slot(ctx)]) {
      HIP_CHECK(hipFuncSetAttribute((const void*)conv1_image_kernel<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, ilds));
      attr_done[cpp_dev_slot(ctx)] = true;
    }
    hipLaunchKernelGGL(conv1_image_kernel<CIN>, dim3(a.n), dim3(CONV_THREADS), ilds, ctx->stream, *ia);
    LAUNCH_CHECK();
    prof_end(ctx, K_CONV1_IMAGE);      // (the caller's bracket: the image launch is timed on its own, the forward kernel after it)
    prof_begin(ctx);
  }
  hipLaunchKernelGGL(conv_fwd_rs16_kernel<CIN>, dim3(((a.a[0].B + 1) / 2) * (a.a[0].nbands > 1 ? 2 : 1), a.n), dim3(CONV_THREADS), 0, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}

// 64 pixels wide (two 32-pixel strips, each with one image border), 3, 6, 9, 12 or 18 channels (one to three 30-k chunks per row: at most 120 weight registers;
// 15 channels -- three chunks AND two copies of the staged row -- spill), pooled output, one whitening table for the batch, FAST precision (two f16 pieces); everything else
// stays on conv_k16.h.
int conv_fwd_rs16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool plain, const ConvArgsN& a, bool* handled) {
  *handled = false;
  static const bool off = cpp_switch_off("CPP_CONV_RS16");
  static const bool off_new = cpp_switch_off("CPP_CONV_RS16_CH");      // (ablation build: only round 5's 18-channel instance)
  const ConvArgs& a0 = a.a[0];
  if (off || plain || !conv_rs16_channels_ok(cin) || (off_new && cin != 18) || ks != 5 || in_mode != IN_F16_WHITEN || a0.W != 64 || a0.H < 16 || (a0.H & 1) || a0.nout != KYO_NO) return 0;
  if (ctx && ctx->precision == CPP_PRECISION_EXACT) return 0;
  for (int i = 0; i < a.n; ++i) {
    if (a.a[i].white_bstride != 0 || a.a[i].wimg == nullptr) return 0;
    if (((uintptr_t)a.a[i].in & 15) || (a.a[i].in_bstride & 7)) return 0;      // (a row segment's 16-byte pieces start where the channel count puts them: 2 .. 8-byte aligned)
  }
  *handled = true;
  if (!ctx) return 0;
  // images the optimiser's launch has already built for these very tables and weights (opt_apply_kernel's rider): nothing to do
  bool fresh = true;
  for (int i = 0; i < a.n; ++i) fresh = fresh && a.a[i].wimg_key != nullptr && a.a[i].wimg_key == a.a[i].scale && a.a[i].wscale == 0.f;
  static const bool no_ride = cpp_switch_off("CPP_RIDE_IMAGE");
  Conv1ImageArgsN ia;
  const bool build = !(fresh && !no_ride);
  if (build) {
    memset(&ia, 0, sizeof(ia)); ia.n = a.n;
    for (int i = 0; i < a.n; ++i)
      ia.a[i] = Conv1ImageArgs{a.a[i].w, a.a[i].bias, a.a[i].scale, a.a[i].shift, a.a[i].wscale, a.a[i].nout,
                               reinterpret_cast<unsigned char*>(const_cast<void*>(a.a[i].wimg)), nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, 0.f, nullptr, nullptr};
  }
  const Conv1ImageArgsN* iap = build ? &ia : nullptr;
  // Two bands of output rows per image when whole images put at most ONE workgroup on a CU (the kernel is built for two waves per SIMD:
  // NAF's two trunks at B = 256 are 256 workgroups): the first band's height r0 is even with r0 - 2 a multiple of the row loop's period
  // (6), as close to H / 2 as that allows (conv_rs16.h; CPP_CONV_BANDS=0 in the ablation build: whole images)
  ConvArgsN ab = a;
  {
    static const bool no_bands = cpp_switch_off("CPP_CONV_BANDS");
    int r0 = 0;
    for (int r = 8; r + 8 <= a0.H; r += 6) if (r0 == 0 || abs(2 * r - a0.H) < abs(2 * r0 - a0.H)) r0 = r;      // r = 8, 14, 20, ...: r - 2 = 6 k
    const int wgs = a.n * ((a0.B + 1) / 2);
    const bool two = !no_bands && r0 != 0 && a0.H >= 32 && wgs <= ctx->num_cus;
    for (int i = 0; i < ab.n; ++i) { ab.a[i].nbands = two ? 2 : 0; ab.a[i].band_rows = two ? r0 : 0; }
  }
  switch (cin) {
    case 3: return conv_fwd_rs16_launch<3>(ctx, ab, iap);
    case 6: return conv_fwd_rs16_launch<6>(ctx, ab, iap);
    case 9: return conv_fwd_rs16_launch<9>(ctx, ab, iap);
    case 12: return conv_fwd_rs16_launch<12>(ctx, ab, iap);
    default: return conv_fwd_rs16_launch<18>(ctx, ab, iap);
  }
}
static_assert(Rs16Geom<18>::REC_BYTES >= Rs16Geom<12>::REC_BYTES && Rs16Geom<18>::REC_BYTES >= Rs16Geom<3>::REC_BYTES, "the largest record");
size_t conv_rs16_image_bytes() { return Rs16Geom<18>::REC_BYTES; }
// would conv1 forward of these networks run on conv_rs16.h (and so read an operand image)?
bool conv_rs16_ok(cpp_ctx* ctx, int cin, int H, int W, int nout) {
  static const bool off = cpp_switch_off("CPP_CONV_RS16") || cpp_switch_off("CPP_CONV_K16") || cpp_switch_off("CPP_CONV_KYO") || cpp_switch_off("CPP_RIDE_IMAGE");
  static const bool off_new = cpp_switch_off("CPP_CONV_RS16_CH");
  return !off && !(ctx && ctx->conv1_f32) && conv_rs16_channels_ok(cin) && !(off_new && cin != 18) && W == 64 && H >= 16 && !(H & 1) && nout == KYO_NO && ctx && ctx->precision != CPP_PRECISION_EXACT;
}

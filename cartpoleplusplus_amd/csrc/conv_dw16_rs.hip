// conv1's dW on the f16 pipes, one wave per unit (conv_dw16_rs.h): actor + critic of a minibatch in one grid, alone or with the next
// minibatch's sample + statistics pass behind it (as conv1_dw_gather.hip)
#include <cstring>
#include "conv_dw16_rs.h"
#include "gather_body.h"

template <int CIN>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dw16_rs_kernel(const ConvArgsN batch, int nbands, int band, int gx, const GatherArgs g) {
  const int ndw = gx * (batch.n / 2);
  if ((int)blockIdx.x < ndw) {
    conv_dw16_rs_body<CIN>(batch, nbands, band, (int)blockIdx.x % gx, 2 * ((int)blockIdx.x / gx));
  } else {                                              // the rider (launched with g.B > 0 only)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* dsh = reinterpret_cast<double*>(lds_raw);
    float* sh = reinterpret_cast<float*>(lds_raw + CPP_MAX_CHANNELS * 16 * 8);
    float* lut = sh + 256 * GATHER_SH;
    const int i = (int)blockIdx.x - ndw;
    gather_stats_body<__half>(g, i % g.B, i / g.B, sh, dsh, lut);
  }
}

template <int CIN>
static int conv_dw16_rs_launch(cpp_ctx* ctx, const ConvArgsN& a, int nbands, int band, int gx, int pairs) {
  constexpr int lds = Dw16RsGeom<CIN>::LDS_BYTES > GATHER_LDS_BYTES ? Dw16RsGeom<CIN>::LDS_BYTES : GATHER_LDS_BYTES;      // (the rider's workgroups share the allocation)
  static bool attr_done[CPP_MAX_DEVICES] = {};
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)conv_dw16_rs_kernel<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  GatherArgs g; memset(&g, 0, sizeof(g));
  int nride = 0;
  if (ctx->ride && !ctx->ride_done && ctx->ride_at_dw && ctx->ride_dtype == 1) { g = *ctx->ride; ctx->ride_done = true; nride = 2 * g.B; }
  hipLaunchKernelGGL(conv_dw16_rs_kernel<CIN>, dim3(gx * pairs + nride), dim3(CONV_THREADS), lds, ctx->stream, a, nbands, band, gx, g);
  LAUNCH_CHECK();
  return 0;
}

int conv_dw16_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool dense, const ConvArgsN& a, int* grid, bool* handled) {
  *handled = false;
  static const bool off = cpp_switch_off("CPP_CONV1_DWRS") || cpp_switch_off("CPP_DW16_PAIR");
  // 9 channels (cfg2; round 6): 4 x 4 tiles = 32 MFMAs per row and wave instead of 56 -- and 43.0 us against conv_dw16.h's 40.4: at 9
  // channels the launch is the waves' staging and operand-read chain (every wave converts and stores its own dY rows: the same
  // ~28 us of it as at 18 channels), not its MFMAs.  Kept as an opt-in of the ablation build (CPP_CONV1_DWRS_CH=1), tested against the kernel it does not replace.
  static const bool on9 = cpp_switch_int("CPP_CONV1_DWRS_CH", 0) != 0;
  const ConvArgs& a0 = a.a[0];
  const bool cin_ok = cin == 18 || (on9 && cin == 9);
  if (off || !ctx || dense || !cin_ok || ks != 5 || in_mode != IN_F16_WHITEN || a0.W != 64 || (a0.H & 3) || a0.H < 16 || a0.nout != KYO_NO) return 0;
  if (f16_exact(ctx) || !conv_dw16_pairable(a)) return 0;
  for (int i = 0; i < a.n; ++i)
    if (a.a[i].white_bstride != 0 || ((uintptr_t)a.a[i].in & 3) || (a.a[i].in_bstride & 1)) return 0;
  // bands: two workgroups (eight waves) per CU in one round
  const int pairs = a.n / 2;
  int nbands = 1;
  while (nbands < 4 && (a0.H / (2 * nbands)) % 2 == 0 && a0.H / (2 * nbands) >= 8 && pairs * a0.B * 2 * nbands <= ctx->num_cus * 2) nbands *= 2;
  const int band = a0.H / nbands;
  const int gx = a0.B * nbands;
  if ((size_t)gx > (size_t)ctx->num_cus * 4) return 0;        // (the partial buffers hold num_cus * 4 partials per network)
  *handled = true;
  *grid = gx;
  switch (cin) {
    case 9: return conv_dw16_rs_launch<9>(ctx, a, nbands, band, gx, pairs);
    default: return conv_dw16_rs_launch<18>(ctx, a, nbands, band, gx, pairs);
  }
}

// conv2 / conv3 forward on the bf16 pipes, row-streaming (conv_fw_rs.h)
#include <cstring>
#include "conv_fw_rs.h"

template <int KS, int TPR, int ORDER>
static int conv_fw_rs_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef FwRsGeom<KS, TPR> G;
  const ConvArgs& a = batch.a[0];
  auto kern = conv_fw_rs_kernel<KS, TPR, ORDER>;
  static bool attr_done[CPP_MAX_DEVICES] = {};
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  hipLaunchKernelGGL(kern, dim3((a.B + G::IPW - 1) / G::IPW, batch.n), dim3(CONV_THREADS), G::LDS_BYTES, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

int conv_fw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, int epi, const ConvArgsN& a, bool* handled) {
  *handled = false;
  static const bool off = cpp_switch_off("CPP_CONV_FWRS") || cpp_switch_off("CPP_CONV_B16");
  const ConvArgs& a0 = a.a[0];
  // (conv2 -- 5x5 -- on this body was measured and is not instantiated: 60 MFMAs per tile row against the ring kernel's 48, and conv3 can
  // no longer ride as conv2's tail: cfg3 44 + 15 us against 54, cfg5 313 against 287 us; profiles/NOTEBOOK_r05.md)
  const bool geo = ks == 3 && (a0.W == 32 || a0.W == 64);
  if (off || !ctx || !geo || in_mode != IN_F32_PLAIN || epi != EPI_RELU_POOL || cin != KYO_NO || a0.nout != KYO_NO || (a0.H & 1) || a0.H < 4) return 0;
  for (int i = 0; i < a.n; ++i)
    if (((uintptr_t)a.a[i].in & 7) || (a.a[i].in_bstride & 1) || a.a[i].n3_w != nullptr || a.a[i].out == nullptr || ((uintptr_t)a.a[i].out & 3)) return 0;      // (buffer stores of 16 bytes take 4-byte aligned addresses: conv3 writes into the flattened activations, 2561 floats per image)
  *handled = true;
  const bool nine = b16_order(ctx) == B16_NINE;
  if (a0.W == 32) return nine ? conv_fw_rs_launch_t<3, 2, B16_NINE>(ctx, a) : conv_fw_rs_launch_t<3, 2, B16_SIX>(ctx, a);
  return nine ? conv_fw_rs_launch_t<3, 4, B16_NINE>(ctx, a) : conv_fw_rs_launch_t<3, 4, B16_SIX>(ctx, a);
}

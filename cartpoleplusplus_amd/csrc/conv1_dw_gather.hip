// conv1 dW (f16 pipes, conv_dw16.h) of minibatch i and the sample + statistics pass of minibatch i + 1 in ONE launch.  The dW kernel
// is bound by the matrix pipes and leaves HBM idle; its 1024 workgroups run as a full round of 768 and a second round that leaves
// two of three slots per CU empty.  The gather's workgroups come last in the grid and fill those slots: 75 MB of HBM reads
// under MFMA work instead of 20 us behind it.  The rider writes the other set of slot arrays (cpp_batch::slot_alt): conv1's
// dW still reads the current minibatch's.  Its LDS is carved out of the dW kernel's dynamic allocation (a static array would
// cost the dW kernel its third workgroup per CU).
#include "conv_dw16.h"
#include "gather_body.h"

typedef Dw16Geom<18, 5, 2> DwgG;
static_assert(DwgG::LDS_BYTES >= GATHER_LDS_BYTES, "the gather's LDS fits the dW kernel's allocation");      // (three pieces: more)

// the same with two networks per dW workgroup (conv_dw16.h, NNET = 2): the dW part is gx * n / 2 workgroups, two per CU; the rider's
// come behind them -- with the per-state sums kept by the store (GatherArgs::slot_stats) they copy 2 C doubles each and are gone
template <int NPCS>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv1_dw_pair_gather_kernel(const ConvArgsN batch, int units_per_img, int band, int gx, const GatherArgs g) {
  const int ndw = gx * (batch.n / 2);
  if ((int)blockIdx.x < ndw) {
    conv_dw16_body<18, 5, 2, false, 2, NPCS>(batch, units_per_img, band, (int)blockIdx.x % gx, 2 * ((int)blockIdx.x / gx), gx);
  } else {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* dsh = reinterpret_cast<double*>(lds_raw);
    float* sh = reinterpret_cast<float*>(lds_raw + CPP_MAX_CHANNELS * 16 * 8);
    float* lut = sh + 256 * GATHER_SH;
    const int i = (int)blockIdx.x - ndw;
    gather_stats_body<__half>(g, i % g.B, i / g.B, sh, dsh, lut);
  }
}

template <int NPCS>
__global__ __launch_bounds__(CONV_THREADS, DW16_WGS) void conv1_dw_gather_kernel(const ConvArgsN batch, int units_per_img, int band, int gx, const GatherArgs g) {
  const int ndw = gx * batch.n;
  if ((int)blockIdx.x < ndw) {
    conv_dw16_body<18, 5, 2, false, 1, NPCS>(batch, units_per_img, band, (int)blockIdx.x % gx, (int)blockIdx.x / gx, gx);
  } else {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* dsh = reinterpret_cast<double*>(lds_raw);
    float* sh = reinterpret_cast<float*>(lds_raw + CPP_MAX_CHANNELS * 16 * 8);
    float* lut = sh + 256 * GATHER_SH;
    const int i = (int)blockIdx.x - ndw;
    gather_stats_body<__half>(g, i % g.B, i / g.B, sh, dsh, lut);
  }
}

template <int NPCS>
static int launch_conv1_dw_gather_t(cpp_ctx* ctx, const ConvArgsN& batch, int upi, int band, int grid, size_t lds_bytes, const GatherArgs& g, bool pair) {
  static bool attr_done[CPP_MAX_DEVICES][2] = {};       // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)][pair ? 1 : 0]) {
    if (pair) HIP_CHECK(hipFuncSetAttribute((const void*)conv1_dw_pair_gather_kernel<NPCS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    else HIP_CHECK(hipFuncSetAttribute((const void*)conv1_dw_gather_kernel<NPCS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)][pair ? 1 : 0] = true;
  }
  if (pair) hipLaunchKernelGGL(conv1_dw_pair_gather_kernel<NPCS>, dim3(grid * (batch.n / 2) + 2 * g.B), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band, grid, g);
  else hipLaunchKernelGGL(conv1_dw_gather_kernel<NPCS>, dim3(grid * batch.n + 2 * g.B), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band, grid, g);
  LAUNCH_CHECK();
  return 0;
}

int launch_conv1_dw_gather(cpp_ctx* ctx, const ConvArgsN& batch, int upi, int band, int grid, size_t lds_bytes, const GatherArgs& g, bool pair, bool exact) {
  return exact ? launch_conv1_dw_gather_t<F16_PIECES_EXACT>(ctx, batch, upi, band, grid, lds_bytes, g, pair)
               : launch_conv1_dw_gather_t<F16_PIECES>(ctx, batch, upi, band, grid, lds_bytes, g, pair);
}

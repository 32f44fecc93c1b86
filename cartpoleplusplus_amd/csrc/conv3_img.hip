// conv3 forward for 16x16 inputs (base_network.py:119-127: 3x3, 10 -> 10 filters, SAME, ReLU, 2x2 max pool) with the whole
// image in LDS.  The row-streaming kernel (conv_kyo.h) spends 0.74 us per row on this layer for 0.25 us of MFMA -- a staged
// row, an operand chain and a workgroup barrier per row -- and 4 us on set-up; here a workgroup fetches its two images and
// the weights in ONE round trip, and every wave then runs its eight rows from LDS without meeting another wave again.
//   wave = (image, upper / lower half); per pair of rows y, y + 1: 23 k-steps of v_mfma_f32_16x16x4_f32 on two independent
//   accumulators (M = the 16 pixels of a row, N = filters, k = (ky, kx, c) in steps of 4), B operands (the weights) held in
//   registers for the whole kernel, A operands one ds_read_b32 per step from the zero-haloed image.
#include "conv3_img.h"

__global__ __launch_bounds__(256) void conv3_img_kernel(const ConvArgsN batch) {
  const ConvArgs& a = batch.a[blockIdx.y];
  __shared__ __attribute__((aligned(16))) float img[C3_IPW * C3_IMGF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int b0 = blockIdx.x * C3_IPW;
  const int nimg = a.B - b0 < C3_IPW ? a.B - b0 : C3_IPW;

  // ---- every global load of the kernel, issued before the first use
  constexpr int NV = C3_IPW * C3_H * C3_H * C3_C / 4;          // 16-byte chunks of the two images: 1280
  constexpr int NVT = (NV + 255) / 256;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 iv[NVT];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((const float*)a.in + (long)b0 * a.in_bstride), 0, (int)(((long)(nimg - 1) * a.in_bstride + C3_H * C3_H * C3_C) * 4), 0x00020000);
#pragma unroll
  for (int n = 0; n < NVT; ++n) {
    const int ch = tid + n * 256;                              // chunk -> (image, 16-byte piece of its 2560 floats)
    const int im = ch / (C3_H * C3_H * C3_C / 4), j = ch - im * (C3_H * C3_H * C3_C / 4);
    iv[n] = (u32x4){0u, 0u, 0u, 0u};
    if (ch < NV && im < nimg) iv[n] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)((long)im * a.in_bstride * 4) + j * 16, 0, 0);
  }
  Conv3Ops ops;
  conv3_load_ops(ops, a.w, a.bias, a.nout, li, lj);

  // ---- the padded images: zeros, then the interiors
  for (int i = tid; i < C3_IPW * C3_IMGF / 4; i += 256) reinterpret_cast<float4*>(img)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NVT; ++n) {
    const int ch = tid + n * 256;
    if (ch < NV) {
      const int im = ch / (C3_H * C3_H * C3_C / 4), j = ch - im * (C3_H * C3_H * C3_C / 4);
      const int e = j * 4;                                     // first float of the chunk: (y, x, c) = (e / 160, (e % 160) / 10, e % 10)
      const int y = e / (C3_H * C3_C), rx = e - y * (C3_H * C3_C);
      float* dst = img + im * C3_IMGF + ((y + 1) * C3_PW + 1) * C3_C + rx;      // a row's 160 floats stay contiguous in the padded row
      dst[0] = __uint_as_float(iv[n].x); dst[1] = __uint_as_float(iv[n].y); dst[2] = __uint_as_float(iv[n].z); dst[3] = __uint_as_float(iv[n].w);
    }
  }
  __syncthreads();

  // ---- one wave per half image: four pairs of rows
  const int im = wave >> 1, half = wave & 1;
  if (im >= nimg) return;
  conv3_img_half(ops, img + im * C3_IMGF, half, a.out + (long)(b0 + im) * a.out_bstride,
                 a.out_amax + (long)(b0 + im) * (C3_H / 2) * (C3_H / 2) * a.nout, a.nout, li, lj);
}

// CPP_CONV3_IMG=0 keeps the row-streaming kernel
bool conv3_img_ok(int cin, int ks, int H, int W, int nout) {
  static const bool off = cpp_switch_off("CPP_CONV3_IMG");
  return !off && cin == C3_C && ks == C3_KS && H == C3_H && W == C3_H && nout >= 1 && nout <= C3_NO;
}

int launch_conv3_img(cpp_ctx* ctx, const ConvArgsN& batch) {
  const ConvArgs& a = batch.a[0];
  for (int i = 0; i < batch.n; ++i)
    if (((uintptr_t)batch.a[i].in & 15) || (batch.a[i].in_bstride & 3)) { cpp_set_error("conv3 image kernel: unaligned input"); return 1; }
  hipLaunchKernelGGL(conv3_img_kernel, dim3((a.B + C3_IPW - 1) / C3_IPW, batch.n), dim3(256), 0, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

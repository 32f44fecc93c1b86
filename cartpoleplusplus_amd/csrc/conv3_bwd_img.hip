// conv3 backward (dW, db and dX) for 16x16 inputs with whole images in LDS -- the backward twin of conv3_img.hip.
//
// base_network.py:119-127: conv3 = 3x3, 10 -> 10 filters, SAME, ReLU, then a 2x2 max pool.  Its backward pass was two row-streaming
// kernels sharing one launch (conv3_bwd_pair.hip: 16-18 us for 0.47 GFLOP -- per row a staged dY row, an operand chain and a
// workgroup barrier, plus 4 us of set-up each).  Here a workgroup fetches, in ONE round trip, everything its two images need (the
// pooled activations of conv2 = conv3's input, the pooled gradient / pooled output / arg-max codes of conv3, the weights), builds
// in LDS the zero-haloed input image and the zero-haloed dense gradient
//     dZ[y][x][o] = dpool[y/2][x/2][o]  if  code[y/2][x/2][o] == 2 (y & 1) + (x & 1)  and  pool[y/2][x/2][o] > 0,  else 0
// and then every wave runs from LDS without meeting another wave until the final reduction:
//   dX[y][x][c] = sum_{ky,kx,o} dZ[y+ky-1][x+kx-1][o] W[2-ky][2-kx][c][o]         wave = a quarter of the image: 4 rows x 23 k-steps
//   dW[ky][kx][c][o] = sum_{y,x} X[y+ky-1][x+kx-1][c] dZ[y][x][o],  db[o] = sum dZ    wave = 64 pixels: 16 k-steps x 6 row tiles;
//                                                                                   row 90 of the A operand is the constant 1 (db)
// all on v_mfma_f32_16x16x4_f32 (exact f32 products).  One partial (900 + 10 floats) per workgroup for conv_dw_reduce_kernel.
#include <cstring>
#include "conv3_img.h"

constexpr int C3B_MROWS = C3_K + 1;                        // 90 weight rows + the bias row
constexpr int C3B_MT = (C3B_MROWS + 15) / 16;              // 6
constexpr int C3B_NPOOL = (C3_H / 2) * (C3_H / 2) * C3_NO; // 640 pooled cells per image
#ifndef C3B_IPW
#define C3B_IPW 1                                         // images per workgroup (1: 512 workgroups for two networks at B = 256, three per CU)
#endif
constexpr int C3B_WPI = 4 / C3B_IPW;                       // waves per image
constexpr int C3B_LDS_FLOATS = 2 * C3B_IPW * C3_IMGF + 4 * C3B_MT * 64 * 4;     // X images, dZ images, the waves' accumulators

__global__ __launch_bounds__(256) void conv3_bwd_img_kernel(const ConvArgsN dxb, const ConvArgsN dwb) {
  const ConvArgs& ax = dxb.a[blockIdx.y];             // dX side: weights, output (gradient w.r.t. conv3's input)
  const ConvArgs& aw = dwb.a[blockIdx.y];             // dW side: conv3's input images, the partial
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xp = lds;                                    // [C3B_IPW][18][18][10] zero-haloed input images
  float* dz = lds + C3B_IPW * C3_IMGF;                 // [C3B_IPW][18][18][10] zero-haloed dense gradient
  float* red = lds + 2 * C3B_IPW * C3_IMGF;            // [4 waves][C3B_MT][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int b0 = blockIdx.x * C3B_IPW;
  const int nimg = aw.B - b0 < C3B_IPW ? aw.B - b0 : C3B_IPW;
  const int nout = aw.nout;                           // conv3's filters (== its input channels here: 10)

  // ---- every global load of the kernel, issued before the first use
  constexpr int NV = C3B_IPW * C3_H * C3_H * C3_C / 4;          // 16-byte chunks of the two input images: 1280
  constexpr int NVT = (NV + 255) / 256;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 iv[NVT];
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((const float*)aw.in + (long)b0 * aw.in_bstride), 0, (int)(((long)(nimg - 1) * aw.in_bstride + C3_H * C3_H * C3_C) * 4), 0x00020000);
#pragma unroll
  for (int n = 0; n < NVT; ++n) {
    const int ch = tid + n * 256;
    const int im = ch / (C3_H * C3_H * C3_C / 4), j = ch - im * (C3_H * C3_H * C3_C / 4);
    iv[n] = (u32x4){0u, 0u, 0u, 0u};
    if (ch < NV && im < nimg) iv[n] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)((long)im * aw.in_bstride * 4) + j * 16, 0, 0);
  }
  constexpr int NPC = (C3B_IPW * C3B_NPOOL + 255) / 256;        // pooled cells per thread: 5
  float gp[NPC], gd[NPC]; int gc[NPC];
#pragma unroll
  for (int n = 0; n < NPC; ++n) {
    const int cell = tid + n * 256;
    const int im = cell / C3B_NPOOL, j = cell - im * C3B_NPOOL;
    gp[n] = 0.f; gd[n] = 0.f; gc[n] = 0;
    if (cell < C3B_IPW * C3B_NPOOL && im < nimg && (j % C3_NO) < nout) {
      const int jo = (j / C3_NO) * nout + (j % C3_NO);          // (the tensors are packed with nout channels)
      gp[n] = aw.dy.pool[(long)(b0 + im) * aw.dy.pool_bstride + jo];
      gd[n] = aw.dy.dpool[(long)(b0 + im) * aw.dy.dpool_bstride + jo];
      gc[n] = aw.dy.amax[(long)(b0 + im) * (C3_H / 2) * (C3_H / 2) * nout + jo];
    }
  }
  // dX's B operands: W'[k = (ky', kx', o)][n = c] = W[2 - ky'][2 - kx'][c][o]; the k-th tap's float offset from a window's corner
  float bw[C3_STEPS]; int offk[C3_STEPS];
#pragma unroll
  for (int st = 0; st < C3_STEPS; ++st) {
    const int k = 4 * st + lj;
    const bool ok = k < C3_K && li < nout;
    const int kc = k < C3_K ? k : 0;
    const int ky = kc / (C3_KS * C3_C), r = kc - ky * (C3_KS * C3_C), kx = r / C3_C, o = r - kx * C3_C;
    bw[st] = ax.w[ok && o < nout ? (((C3_KS - 1 - ky) * C3_KS + (C3_KS - 1 - kx)) * nout + li) * nout + o : 0];
    if (!ok || o >= nout) bw[st] = 0.f;
    offk[st] = (ky * C3_PW) * C3_C + r;
  }

  // ---- zero-haloed images: zeros, then the input interiors and the routed gradient
  for (int i = tid; i < 2 * C3B_IPW * C3_IMGF / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NVT; ++n) {
    const int ch = tid + n * 256;
    if (ch < NV) {
      const int im = ch / (C3_H * C3_H * C3_C / 4), j = ch - im * (C3_H * C3_H * C3_C / 4);
      const int e = j * 4;
      const int y = e / (C3_H * C3_C), rx = e - y * (C3_H * C3_C);
      float* dst = xp + im * C3_IMGF + ((y + 1) * C3_PW + 1) * C3_C + rx;
      dst[0] = __uint_as_float(iv[n].x); dst[1] = __uint_as_float(iv[n].y); dst[2] = __uint_as_float(iv[n].z); dst[3] = __uint_as_float(iv[n].w);
    }
  }
#pragma unroll
  for (int n = 0; n < NPC; ++n) {
    const int cell = tid + n * 256;
    if (cell < C3B_IPW * C3B_NPOOL) {
      const int im = cell / C3B_NPOOL, j = cell - im * C3B_NPOOL;
      const int o = j % C3_NO, pc = j / C3_NO, py = pc / (C3_H / 2), px = pc - py * (C3_H / 2);
      const float g = gp[n] > 0.f ? gd[n] : 0.f;
      const int y = 2 * py + (gc[n] >> 1), x = 2 * px + (gc[n] & 1);
      dz[im * C3_IMGF + ((y + 1) * C3_PW + (x + 1)) * C3_C + o] = g;
    }
  }
  __syncthreads();

  const int im = wave / C3B_WPI, half = wave % C3B_WPI;      // `half`: which 1 / C3B_WPI of the image's rows
  constexpr int RPW = C3_H / C3B_WPI;                        // rows per wave
  // ---- dX: rows RPW half .. RPW half + RPW - 1 of image im (plain rows out)
  if (im < nimg) {
    const float* base = dz + im * C3_IMGF + li * C3_C;         // window corner of pixel x = li in padded row 0
    float* out = ax.out + (long)(b0 + im) * ax.out_bstride;
#pragma unroll 1
    for (int pp = 0; pp < RPW / 2; ++pp) {
      const int y = RPW * half + 2 * pp;
      const float* r0 = base + y * C3_PW * C3_C;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < C3_STEPS; ++st) {
        const float a0 = r0[offk[st]], a1 = r0[offk[st] + C3_PW * C3_C];
        acc0 = C3_MFMA16(a0, bw[st], acc0);
        acc1 = C3_MFMA16(a1, bw[st], acc1);
      }
      if (li < nout) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          out[(y * C3_H + 4 * lj + r) * nout + li] = acc0[r];
          out[((y + 1) * C3_H + 4 * lj + r) * nout + li] = acc1[r];
        }
      }
    }
  }

  // ---- dW / db: this wave's 128 pixels.  A[m][pixel] = X[y + ky - 1][x + kx - 1][c] (m = (ky, kx, c) < 90), 1 (m == 90), 0 above;
  // B[pixel][o] = dZ[y][x][o]
  f32x4 acc[C3B_MT];
#pragma unroll
  for (int mt = 0; mt < C3B_MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int offm[C3B_MT]; bool mreal[C3B_MT], mone[C3B_MT];
#pragma unroll
  for (int mt = 0; mt < C3B_MT; ++mt) {
    const int m = 16 * mt + li;
    mreal[mt] = m < C3_K; mone[mt] = m == C3_K;
    const int mc = mreal[mt] ? m : 0;
    const int ky = mc / (C3_KS * C3_C), r = mc - ky * (C3_KS * C3_C);
    offm[mt] = (ky * C3_PW) * C3_C + r;                       // padded-image offset of tap (ky, kx, c) from the window corner
  }
  if (im < nimg) {
    const float* xim = xp + im * C3_IMGF;
    const float* zim = dz + im * C3_IMGF + (C3_PW + 1) * C3_C + (li < nout ? li : 0);
#pragma unroll 4
    for (int st = 0; st < RPW * C3_H / 4; ++st) {
      const int p = RPW * C3_H * half + 4 * st + lj, y = p >> 4, x = p & 15;
      const int corner = (y * C3_PW + x) * C3_C;
      const float b = li < nout ? zim[corner] : 0.f;
#pragma unroll
      for (int mt = 0; mt < C3B_MT; ++mt) {
        float av = xim[corner + offm[mt]];
        av = mreal[mt] ? av : (mone[mt] ? 1.f : 0.f);
        acc[mt] = C3_MFMA16(av, b, acc[mt]);
      }
    }
  }
  // ---- the four waves' accumulators -> one partial, added in wave order
#pragma unroll
  for (int mt = 0; mt < C3B_MT; ++mt)
    *reinterpret_cast<f32x4*>(red + ((wave * C3B_MT + mt) * 64 + lane) * 4) = acc[mt];
  __syncthreads();
  float* part = aw.partial + (long)blockIdx.x * aw.pstride;
  const int nw = C3_KS * C3_KS * nout * nout;
  for (int e = tid; e < C3B_MT * 256; e += 256) {              // e = (mt, lane', r): D[m = 16 mt + 4 lj' + r][n = li']
    const int mt = e >> 8, ln = (e >> 2) & 63, r = e & 3;
    const int m = 16 * mt + 4 * (ln >> 4) + r, o = ln & 15;
    if (m <= C3_K && o < nout) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * C3B_MT + mt) * 64 + ln) * 4 + r];
      if (m < C3_K) {
        const int ky = m / (C3_KS * C3_C), rr = m - ky * (C3_KS * C3_C), kx = rr / C3_C, c = rr - kx * C3_C;
        if (c < nout) part[((ky * C3_KS + kx) * nout + c) * nout + o] = s;
      } else {
        part[nw + o] = s;
      }
    }
  }
}

// MEASURED EQUAL to the two row kernels in one launch (conv3_bwd_pair.hip): 18.0 vs 17.8 us per launch at cfg3 with one image per
// workgroup, 20.4 with two (profiles/experiments/r03_pairs.txt) -- the launch is a chain of fixed latencies either way.  Kept as an
// ablation: on only with CPP_CONV3_BWD_IMG=1 in the ablation build.
bool conv3_bwd_img_ok(int cin, int ks, int H, int W, int nout) {
  static const bool on = cpp_switch_int("CPP_CONV3_BWD_IMG", 0) != 0;
  return on && cin == C3_C && ks == C3_KS && H == C3_H && W == C3_H && nout == C3_NO;
}

// dxb / dwb: the same networks' dX and dW descriptors (conv_dx_args / conv_dw_args).  *grid = partials per network.
int launch_conv3_bwd_img(cpp_ctx* ctx, const ConvArgsN& dxb, const ConvArgsN& dwb, int* grid) {
  const ConvArgs& a = dwb.a[0];
  for (int i = 0; i < dwb.n; ++i)
    if (((uintptr_t)dwb.a[i].in & 15) || (dwb.a[i].in_bstride & 3)) { cpp_set_error("conv3 backward image kernel: unaligned input"); return 1; }
  const size_t lds = (size_t)C3B_LDS_FLOATS * 4;
  static bool attr_done[CPP_MAX_DEVICES] = {};
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)conv3_bwd_img_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int gx = (a.B + C3B_IPW - 1) / C3B_IPW;
  prof_begin(ctx);
  hipLaunchKernelGGL(conv3_bwd_img_kernel, dim3(gx, dwb.n), dim3(256), lds, ctx->stream, dxb, dwb);
  LAUNCH_CHECK();
  prof_end(ctx, K_CONV3_BWD);
  *grid = gx;
  return 0;
}

// Forward convolution, "(ky, o)-column" formulation for v_mfma_f32_16x16x4_f32 (exact f32).
//
// conv_fwd_kernel (conv_impl.h) puts only the 10 filters of slim.conv2d (base_network.py:103-123) in the MFMA's
// 16 columns, so 6 of 16 columns multiply zeros.  Here the vertical taps sit next to the filters:
//
//     D_q[x, (p, o)] = sum_{kx, c} in[q, x + kx - P, c] * W[ky(q, p), kx, c, o]          N = KS * 10 -> 50 of 64 columns
//
// for ONE input row q; column block p holds the running sum of the output row y with y mod KS == p that row q
// contributes to, i.e. ky(q, p) = (q + P - p) mod KS.  A wave walks down the image, one input row per step, and
// keeps the KS in-flight output rows of its 16*XT columns in the accumulators; the weights rotate (the B operand is
// read from LDS with a per-lane ky offset) instead of the accumulators, so no cross-lane traffic is needed until an
// output row is complete: after step q row y = q - P is final, its block is taken out (bias, x half of the 2x2
// max-pool in registers, the y half through a small wave-private LDS buffer, ReLU, arg-max code), zeroed and reused
// for row y + KS.  Zero-padding rows are skipped, so there is no halo recomputation at all:
//     MFMAs per 16 output pixels:  KS * ceil(KS*CIN/4) (conv_fwd_kernel)  ->  ceil(KS*CIN/4) * ceil(KS*10/16)
//     conv1 (5x5x18): 115 -> 92;  conv2 (5x5x10): 65 -> 52;  conv3 (3x3x10): 24 -> 16.
//
// Workgroup = 4 waves = IPW images x (4 / IPW) column strips of 16*XT; each wave owns one strip of one image for
// the whole image height.  An input row is used by exactly one step, so LDS holds just two rows (double buffer):
// row q+1 is written while row q is multiplied, the global loads for row q+2 are already in flight.
// The K order inside a row is permuted so that a lane's A (and B) operands of 4 consecutive k-steps are contiguous:
// one ds_read2_b64 / ds_read_b128 per 4 MFMAs instead of one ds_read_b32 per MFMA.
#pragma once
#include "conv_impl.h"

constexpr int KYO_NO = 10;                         // filters per layer (base_network.py:103,111,119)

template <int CIN, int KS, int XT, int IPW>
struct KyoGeom {
  static constexpr int P = KS / 2;
  static constexpr int NT = (KS * KYO_NO + 15) / 16;        // N tiles over columns (p, o)
  static constexpr int STRIPS = 4 / IPW, SW = 16 * XT, WPAD = STRIPS * SW;
  static constexpr int KROW = KS * CIN;                     // k = (kx, c) of one input row
  static constexpr int NG = KROW / 16, REM = KROW % 16, RS = (REM + 3) / 4;
  static constexpr int NGT = NG + (RS > 0 ? 1 : 0);         // groups of up to 4 k-steps
  static constexpr int KSTEPS = 4 * NG + RS;
  static constexpr int FP = (4 - (P * CIN) % 4) % 4;        // front pad: image column 0 lands 16-byte aligned
  static constexpr int ROWF = ((FP + (WPAD + KS - 1) * CIN + 8) + 3) & ~3;   // floats per staged row (+ k over-read)
  static constexpr int WLF = KS * NGT * 4 * KYO_NO * 4;     // weight floats in LDS: [ky][group][lj][o][step]
  static constexpr int RING = 3;                            // staged rows in flight: row q is multiplied, q+1 is
                                                            // already visible (its first operands prefetch across the
                                                            // barrier), q+2 is being written
  static constexpr int EF = 2 * 8 * XT * KYO_NO * 2;        // per wave: (value, code) of the two rows of a pool pair
  static constexpr int LDS_FLOATS = WLF + RING * IPW * ROWF + 4 * EF;
  // steps of group g and the k index (within the row) of lane group lj at step s
  static __host__ __device__ constexpr int steps(int g) { return g < NG ? 4 : RS; }
  static __host__ __device__ constexpr int kidx(int g, int s, int lj) {
    return (g < NG || RS == 4) ? 16 * g + 4 * lj + s
         : (RS == 3) ? (s < 2 ? 16 * g + 2 * lj + s : 16 * g + 8 + lj)
         : (RS == 2) ? 16 * g + 2 * lj + s
         : 16 * g + lj;
  }
};

template <int CIN, int KS, int XT, int IPW, int IN_MODE>
__global__ __launch_bounds__(CONV_THREADS, (XT == 1 ? 4 : 2)) void conv_fwd_kyo_kernel(const ConvArgsN batch) {
  typedef KyoGeom<CIN, KS, XT, IPW> G;
  typedef typename StageType<IN_MODE>::type ST;
  constexpr bool WHITEN = (IN_MODE == IN_F16_WHITEN || IN_MODE == IN_F32_WHITEN);
  constexpr int P = G::P, NT = G::NT, NGT = G::NGT, ROWF = G::ROWF, NO = KYO_NO;
  constexpr int EPC = ChunkOps<ST>::EPC;
  constexpr bool A64 = (CIN % 2 == 0) && (G::FP % 2 == 0);      // A operand pairs are 8-byte aligned in LDS
  constexpr int RING = G::RING, RSET = IPW * ROWF;              // row buffers in flight, floats per buffer
#ifdef KYO_CLOCK_PROBE
  const unsigned long long pe = __builtin_amdgcn_s_memrealtime();
#endif
  const ConvArgs& a = batch.a[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                                   // [KS][NGT][4][NO][4]
  float* rows = lds + G::WLF;                        // [RING][IPW][ROWF]
  float2* ebuf = reinterpret_cast<float2*>(rows + RING * RSET);   // [4 waves][2 parities][8*XT][NO] (value, code)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int img = wave / G::STRIPS, strip = wave % G::STRIPS;
  const int b0 = blockIdx.x * IPW;
  const int bimg = b0 + img;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;

  // ---- one-time setup: zero the row buffers (padding columns stay zero), weights into the permuted LDS layout
  for (int i = tid; i < RING * RSET; i += CONV_THREADS) rows[i] = 0.f;
  {
    constexpr int NW = (G::WLF + CONV_THREADS - 1) / CONV_THREADS;
    float wv[NW];
#pragma unroll
    for (int n = 0; n < NW; ++n) {                   // branch-free: all loads in flight before the first LDS write
      const int i = tid + n * CONV_THREADS;
      const int s = i & 3, o = (i >> 2) % NO, r = (i >> 2) / NO;
      const int l = r & 3, g = (r >> 2) % NGT, ky = (r >> 2) / NGT;
      const int k = G::kidx(g, s, l);
      const bool ok = i < G::WLF && s < G::steps(g) && o < nout && k < G::KROW;
      const float v = a.w[ok ? (ky * G::KROW + k) * nout + o : 0];
      wv[n] = ok ? v : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NW; ++n)
      if (tid + n * CONV_THREADS < G::WLF) wl[tid + n * CONV_THREADS] = wv[n];
  }

  // ---- row staging: chunk ch -> (image, 16-byte chunk of the row); a thread's chunks are the same for every row,
  // so the source pointer, the LDS destination and the per-element whitening constants are fixed up front
  const int cpr = (W * CIN) / EPC;                   // chunks per image row (W*CIN % EPC == 0 checked by the host)
  constexpr int NVMAX = (IPW * G::WPAD * CIN / EPC + CONV_THREADS - 1) / CONV_THREADS;
  uint4 sv[NVMAX];
  const ST* ssrc[NVMAX];
  int sdst[NVMAX];
  float2 swh[WHITEN ? NVMAX : 1][EPC];
#pragma unroll
  for (int i = 0; i < NVMAX; ++i) {
    const int ch = tid + CONV_THREADS * i;
    const int im = ch / cpr, j = ch - im * cpr;
    const bool act = im < IPW && b0 + im < a.B;
    ssrc[i] = act ? (const ST*)a.in + (long)(b0 + im) * a.in_bstride + j * EPC : nullptr;
    sdst[i] = im * ROWF + G::FP + P * CIN + j * EPC;
    if (WHITEN) {
      int c = (j * EPC) % CIN;
#pragma unroll
      for (int k = 0; k < EPC; ++k) {
        swh[i][k] = make_float2(a.scale[c], a.shift[c]);
        c = (c + 1 == CIN) ? 0 : c + 1;
      }
    }
  }
  auto stage_load = [&](int y) {
#pragma unroll
    for (int i = 0; i < NVMAX; ++i)
      if (ssrc[i]) sv[i] = *reinterpret_cast<const uint4*>(ssrc[i] + (long)y * W * CIN);
  };
  auto stage_store = [&](float* dstrows) {
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      if (ssrc[i]) {
        float x[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) {
          x[k] = ChunkOps<ST>::get(sv[i], k);
          if (WHITEN) x[k] = x[k] * swh[i][k].x + swh[i][k].y;
        }
        float* dst = dstrows + sdst[i];
#pragma unroll
        for (int k = 0; k < EPC; k += 4)
          *reinterpret_cast<float4*>(dst + k) = make_float4(x[k], x[k + 1], x[k + 2], x[k + 3]);
      }
    }
  };

  // ---- per-lane column bookkeeping: column j = 16 t + li = p * NO + o of N tile t.  Consecutive tiles are 16
  // columns apart (> NO), so for a given block p a lane holds at most ONE tile with a column of that block.
  int pt[NT], wofs[NT], eofs[NT];
  float biast[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = 16 * t + li;
    const bool valid = j < KS * NO && (j % NO) < nout;
    const int o = valid ? j % NO : 0;
    pt[t] = valid ? j / NO : -1;
    biast[t] = a.bias[o];
    eofs[t] = (lj * 2) * NO + o;                               // slot of (pooled x = 2 lj, o) in the pool-pair buffer
    const int p = valid ? pt[t] : 0;
    const int ky0 = (P - p + KS) % KS;                         // ky at q = 0
    wofs[t] = ((ky0 * NGT * 4 + lj) * NO + o) * 4;              // float offset of (ky, group 0, lj, o, step 0)
  }
  constexpr int KYSTRIDE = NGT * 4 * NO * 4, GSTRIDE = 4 * NO * 4;

  // ---- pooled-row writer: entry idx = xl * nout + o of this wave's 8*XT pooled columns, two entries per lane at most
  constexpr int NC = (8 * XT * NO + 63) / 64;
  int ce[NC], coe[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = lane + 64 * i;
    const int xl = idx / nout, o = idx - xl * nout;
    const int px = ((strip * G::SW) >> 1) + xl;
    const bool ok = idx < 8 * XT * nout && px < Wp && bimg < a.B;
    ce[i] = ok ? xl * NO + o : -1;
    coe[i] = px * nout + o;
  }

  f32x4 acc[XT][NT];
#pragma unroll
  for (int m = 0; m < XT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  __syncthreads();                                   // zeroed row buffers visible before the first rows are written
  for (int r = 0; r < RING - 1; ++r) {
    if (r < H) { stage_load(r); stage_store(rows + r * RSET); }
  }
  if (RING - 1 < H) stage_load(RING - 1);
  __syncthreads();

  float2* ev = ebuf + wave * (2 * 8 * XT * NO);      // [2 parities][8*XT][NO]
  const int xcol0 = strip * G::SW;                   // first image column of this wave's strip
  int pdone = (KS - P) % KS;                         // block of row y = q - P
  int rcur = 0;                                      // ring slot of row q

  // operands of one group of k-steps; two sets so that the loads of group g+1 are in flight under the MFMAs of g
  float av[2][XT][4];
  f32x4 bv[2][NT];
  auto load_ops = [&](int g, int set, const float* ab) {
#pragma unroll
    for (int m = 0; m < XT; ++m) {
      const float* ap = ab + m * 16 * CIN;
#ifdef KYO_ABL_NOLDSA
      av[set][m][0] = av[set][m][1] = av[set][m][2] = av[set][m][3] = biast[0];
      if (false) {} else
#endif
      if (A64 && (g < G::NG || G::RS == 4)) {
        const float2 u0 = *reinterpret_cast<const float2*>(ap + 16 * g + 4 * lj);
        const float2 u1 = *reinterpret_cast<const float2*>(ap + 16 * g + 4 * lj + 2);
        av[set][m][0] = u0.x; av[set][m][1] = u0.y; av[set][m][2] = u1.x; av[set][m][3] = u1.y;
      } else if (A64 && G::RS >= 2) {
        const float2 u0 = *reinterpret_cast<const float2*>(ap + 16 * g + 2 * lj);
        av[set][m][0] = u0.x; av[set][m][1] = u0.y;
        av[set][m][2] = (G::RS == 3) ? ap[16 * g + 8 + lj] : 0.f; av[set][m][3] = 0.f;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) av[set][m][s] = s < G::steps(g) ? ap[G::kidx(g, s, lj)] : 0.f;
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#ifdef KYO_ABL_NOLDSB
      bv[set][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};
#else
      bv[set][t] = *reinterpret_cast<const f32x4*>(wl + wofs[t] + g * GSTRIDE);
#endif
    }
  };
  const int aofs = img * ROWF + G::FP + (xcol0 + li) * CIN;    // operand k = (kx, c) of output column x starts at x * CIN
  load_ops(0, 0, rows + aofs);

#ifdef KYO_CLOCK_PROBE
  const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int q = 0; q < H + P; ++q) {
    const int rnext = rcur + 1 == RING ? 0 : rcur + 1;
#ifndef KYO_ABL_NOSTAGE
    {                                                // row q + RING - 1 -> the slot row q - 1 has just left
      const int rw = rcur == 0 ? RING - 1 : rcur - 1;
      if (q + RING - 1 < H) stage_store(rows + rw * RSET);
      if (q + RING < H) stage_load(q + RING);
    }
#endif

    if (q < H) {
      const float* ab = rows + rcur * RSET + aofs;
#ifdef KYO_PRIO
      __builtin_amdgcn_s_setprio(KYO_PRIO);
#endif
#pragma unroll
      for (int g = 0; g < NGT; ++g) {
        if (g + 1 < NGT) load_ops(g + 1, (g + 1) & 1, ab);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s < G::steps(g)) {
#pragma unroll
            for (int m = 0; m < XT; ++m)
#pragma unroll
              for (int t = 0; t < NT; ++t) acc[m][t] = MFMA16(av[g & 1][m][s], bv[g & 1][t][s], acc[m][t]);
          }
        }
      }
#ifdef KYO_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    // rotate the weights: ky of every block advances by one
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      wofs[t] += KYSTRIDE;
      if (wofs[t] >= KS * KYSTRIDE) wofs[t] -= KS * KYSTRIDE;
    }
    // group 0 of the next row (made visible by the previous barrier) loads under the epilogue
    if (q + 1 < H) {
      load_ops(0, NGT & 1, rows + rnext * RSET + aofs);
      if (NGT & 1) {
#pragma unroll
        for (int m = 0; m < XT; ++m)
#pragma unroll
          for (int s = 0; s < 4; ++s) av[0][m][s] = av[1][m][s];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[0][t] = bv[1][t];
      }
    }

    // ---- output row y = q - P is complete (rows y < 0 do not exist: their block only needs clearing): take the
    // block out of the accumulators with selects (no divergence), x half of the max-pool in registers
#ifdef KYO_ABL_NOEPI
    const int y = (q == H + P - 1) ? q - P : -1;
#else
    const int y = q - P;
#endif
    const int par = y & 1;
    {
      f32x4 z[XT];
      float bsel = 0.f; int esel = 0; bool mine = false;
#pragma unroll
      for (int m = 0; m < XT; ++m) z[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool hit = pt[t] == pdone;
        mine = mine || hit;
        bsel = hit ? biast[t] : bsel;
        esel = hit ? eofs[t] : esel;
#pragma unroll
        for (int m = 0; m < XT; ++m) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            z[m][r] = hit ? acc[m][t][r] : z[m][r];
            acc[m][t][r] = hit ? 0.f : acc[m][t][r];
          }
        }
      }
      if (mine && y >= 0) {
#pragma unroll
        for (int m = 0; m < XT; ++m) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float z0 = z[m][2 * h] + bsel, z1 = z[m][2 * h + 1] + bsel;
            ev[par * (8 * XT * NO) + (m * 8 + h) * NO + esel] =
                make_float2(z1 > z0 ? z1 : z0, __int_as_float(z1 > z0 ? 1 : 0));
          }
        }
      }
    }
    pdone = pdone + 1 == KS ? 0 : pdone + 1;
    if (y >= 0 && par == 1 && (y >> 1) < Hp) {         // wave-uniform: both rows of a pool pair are in the buffer
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long orow = (long)(y >> 1) * Wp * nout;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        if (ce[i] >= 0) {
          const float2 top = ev[ce[i]], bot = ev[8 * XT * NO + ce[i]];
          const bool lower = bot.x > top.x;
          const float mx = lower ? bot.x : top.x;
          const int code = lower ? 2 + __float_as_int(bot.y) : __float_as_int(top.y);
#ifdef KYO_ABL_NOSTORE
          if (mx == 123.456f)
#endif
          a.out[(long)bimg * a.out_bstride + orow + coe[i]] = fmaxf(mx, 0.f);
#if defined(KYO_ABL_NOSTORE) || defined(KYO_ABL_NOCODE)
          if (mx == 123.456f)
#endif
          a.out_amax[(long)bimg * Hp * Wp * nout + orow + coe[i]] = (uint8_t)code;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    rcur = rnext;
#ifndef KYO_ABL_NOBAR
    __syncthreads();
#endif
  }
#ifdef KYO_CLOCK_PROBE
  if (tid == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) {
    const unsigned long long pc1 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
    printf("KYOCLK block %d,%d: %llu core cycles in loop; ref ticks: entry %llu loop %llu end %llu\n", (int)blockIdx.x, (int)blockIdx.y,
           pc1 - pc0, pe, pr0, pr1);
  }
#endif
}

template <int CIN, int KS, int XT, int IPW, int IN_MODE>
static inline int conv_fwd_kyo_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef KyoGeom<CIN, KS, XT, IPW> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
  auto kern = conv_fwd_kyo_kernel<CIN, KS, XT, IPW, IN_MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done = true;
  }
  const int grid = (a.B + IPW - 1) / IPW;
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

int conv_fwd_kyo_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled);
int conv_fwd_kyo_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled);

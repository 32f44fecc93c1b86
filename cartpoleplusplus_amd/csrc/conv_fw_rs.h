// conv3 forward (written for both 5x5 and 3x3; the 3x3 instances ship) on the bf16 matrix pipes, ROW-STREAMING with the weights resident in registers ("fwrs"): v_mfma_f32_16x16x32_bf16.
//
//   z[b, y, x, o] = bias_o + sum_{ky, kx, c} in[b, y + ky - P, x + kx - P, c] W[ky][kx][c][o];  out = max-pool 2x2 (relu(z)), arg-max code
//   (base_network.py:111-123: the second and third conv + pool of the trunk)
//
// conv_dx_rs.h's formulation with the weights as they are (no flip), the input row split into three bf16 planes while it is staged (an
// f32 IS the sum of three bf16 numbers; six piece products, CPP_PRECISION_EXACT: nine -- conv2 forward's contract on conv_k16.h), the
// bias as the accumulator sets' restart value, and conv_rs16.h's epilogue: ReLU + 2x2 pool + arg-max code in registers (the x neighbour
// through a DPP quad permutation, the y neighbour = the previous output row's registers), the pooled row through LDS into row layout,
// one 16-byte + one 4-byte store per lane.  A wave owns a 16-pixel tile of an image for all rows; no barrier in the row loop.
#pragma once
#include <type_traits>
#include "conv_k16.h"

typedef unsigned fwrs_u32x2 __attribute__((ext_vector_type(2)));

template <int KS_, int TPR>                                  // kernel size; tiles per row: W = 16 TPR
struct FwRsGeom {
  static constexpr int KS = KS_, P = KS / 2, CH = KYO_NO, NCH = (KS * CH + 31) / 32, NSET = KS + 1;
  static constexpr int UNR = KS == 5 ? 6 : 4;                // steps per unrolled block: a multiple of NSET (sets) and 2 (slots)
  static constexpr int W = 16 * TPR, IPW = 4 / TPR;           // images per workgroup
  static constexpr int NW = KS * KS * CH * CH;
  static constexpr int AIMG_BYTES = KS * NCH * 3 * 1024;      // the A operands: one KB (64 lanes x 16 bytes) per (ky, chunk, piece)
  static constexpr int WPX = 16 + 2 * P;                      // pixels of a staged row
  static constexpr int PLB = ((2 * CH * 15 + 64 * (NCH - 1) + 64 + 15) & ~15) + 16;      // a staged plane (+ the over-read of the last chunk's windows)
  static constexpr int SLOTB = 3 * PLB;
  static constexpr int TRB = 320 + 80 + 16;                   // a pooled row of the tile on its way out: 8 px x 10 ch f32, 80 code bytes, a dump slot
  static constexpr int WVB = 2 * SLOTB + TRB;
  static constexpr int LDS_BYTES = AIMG_BYTES + 4 * WVB;
  static_assert((KS == 5 || KS == 3) && 4 % TPR == 0 && WVB % 16 == 0 && PLB >= 2 * CH * WPX + 8, "geometry");
};

__device__ __forceinline__ float fwrs_swap1(float v) {      // the value of lane ^ 1
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
}

template <int KSZ, int TPR, int ORDER>
__device__ __forceinline__ void conv_fw_rs_body(const ConvArgsN& batch, const int bx, const int by) {
  typedef FwRsGeom<KSZ, TPR> G;
  constexpr int KS = G::KS, P = G::P, CH = G::CH, NCH = G::NCH, NSET = G::NSET, UNR = G::UNR, W = G::W, Wp = W / 2;
  constexpr int PLB = G::PLB, SLOTB = G::SLOTB;
  constexpr unsigned BIG = 0x08000000u;
  const ConvArgs& a = batch.a[by];
  extern __shared__ __attribute__((aligned(16))) unsigned char fwrs_lds[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lj = lane >> 4;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int simg = swave / TPR, stile = swave % TPR;
  const int sb = bx * G::IPW + simg;
  const bool work = sb < a.B;
  const int H = a.H, Hp = H >> 1;
  unsigned char* aimg = fwrs_lds;
  unsigned char* wvb = fwrs_lds + G::AIMG_BYTES + swave * G::WVB;

  // ---- input rows: lane owns channel pairs (wx, 2 cp) of the tile's window (its 16 pixels and P on each side), idx = lane + 64 i = 5 wx + cp
  constexpr int NXV = (G::WPX * 5 + 63) / 64;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((const float*)a.in + (long)(work ? sb : 0) * a.in_bstride), 0, work ? H * W * CH * 4 : 0, 0x00020000);
  unsigned xvo[NXV]; uint32_t xdst[NXV];
#pragma unroll
  for (int i = 0; i < NXV; ++i) {
    const int idx = lane + 64 * i, wx = idx / 5, cp = idx - 5 * wx;
    const int x = stile * 16 - P + wx;
    const bool on = idx < G::WPX * 5;
    xvo[i] = (on && x >= 0 && x < W) ? (unsigned)((x * CH + 2 * cp) * 4) : BIG;
    xdst[i] = keep_in_vgpr(lds_addr(wvb + (on ? 2 * CH * wx + 4 * cp : PLB - 8)));      // (idle lanes: zeros into the plane's tail)
  }
  f32x2 xraw[2][NXV];                                         // two rows in flight
  auto x_load = [&](const int buf, const int q) __attribute__((always_inline)) {      // (the row offset in the VGPR: the range check does not see soffset)
    const unsigned ro = q < H ? (unsigned)(q * (W * CH * 4)) : BIG;
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const fwrs_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)(xvo[i] + ro), 0, 0);
      xraw[buf][i] = (f32x2){__uint_as_float(v.x), __uint_as_float(v.y)};
    }
  };
  auto x_store = [&](const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      unsigned short h0, m0, l0, h1, m1, l1;
      k16_split3(xraw[buf][i][0], h0, m0, l0); k16_split3(xraw[buf][i][1], h1, m1, l1);
      lds_store(xdst[i], slot * SLOTB, (unsigned)h0 | ((unsigned)h1 << 16));
      lds_store(xdst[i], slot * SLOTB + PLB, (unsigned)m0 | ((unsigned)m1 << 16));
      lds_store(xdst[i], slot * SLOTB + 2 * PLB, (unsigned)l0 | ((unsigned)l1 << 16));
    }
  };
  x_load(0, 0); x_load(1, 1);                                 // (on their way while the workgroup builds the operands)

  // ---- the A operands W[ky][k = (kx, c)][o], three bf16 pieces each: built once per workgroup (its waves serve one network)
  {
    constexpr int NWT = (G::NW + CONV_THREADS - 1) / CONV_THREADS;
    float wr[NWT];
#pragma unroll
    for (int n = 0; n < NWT; ++n) { const int i = tid + n * CONV_THREADS; wr[n] = a.w[i < G::NW ? i : 0]; }
    for (int i = tid; i < G::AIMG_BYTES / 16; i += CONV_THREADS) reinterpret_cast<k16_u32x4*>(aimg)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
    for (int i = lane; i < G::WVB / 16; i += 64) reinterpret_cast<k16_u32x4*>(wvb)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
      const int i = tid + n * CONV_THREADS;                   // W[ky][kx][c][o], o fastest
      if (i < G::NW) {
        const int o = i % CH, c = (i / CH) % CH, t = i / (CH * CH), kx = t % KS, ky = t / KS;
        const int k = kx * CH + c, ch = k >> 5, lg = (k >> 3) & 3, e = k & 7;
        unsigned short h, m, l;
        k16_split3(a.wscale != 0.f ? wr[n] * a.wscale : wr[n], h, m, l);
        unsigned short* d = reinterpret_cast<unsigned short*>(aimg + ((ky * NCH + ch) * 3) * 1024 + (lg * 16 + o) * 16 + e * 2);
        d[0] = h; d[512] = m; d[1024] = l;
      }
    }
  }
  __syncthreads();
  if (!work) return;                                          // (wave-uniform; no barrier below)
  k16_u32x4 wv[KS][NCH][3];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        wv[ky][ch][pc] = *reinterpret_cast<const k16_u32x4*>(aimg + ((ky * NCH + ch) * 3 + pc) * 1024 + lane * 16);
  // the sets' restart value: the bias of this lane's four filters
  f32x4 biasv;
#pragma unroll
  for (int r = 0; r < 4; ++r) biasv[r] = (a.bias && 4 * lj + r < a.nout) ? a.bias[4 * lj + r] : 0.f;

  // operand windows: pixel li of the tile, taps kx -> staged pixels li .. li + KS - 1; 8 consecutive k = 16 bytes at 20 li + 64 ch + 16 lj
  const uint32_t xrd = keep_in_vgpr(lds_addr(wvb + 2 * CH * li + 16 * lj));
  k16_u32x4 xb[NCH][3];
  auto read_x = [&](const int ch, const int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      unsigned t[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = lds_load<unsigned>(xrd, slot * SLOTB + p * PLB + 64 * ch + 4 * i);
      xb[ch][p] = (k16_u32x4){t[0], t[1], t[2], t[3]};
    }
  };

  // ---- outputs.  Accumulator layout: lane (li, lj) holds filters 4 lj .. 4 lj + 3 of pixel li; the even lane of a pair ends up with the
  // pooled pixel li / 2 and writes it into the wave's LDS row [8 px][10 ch] (+ codes); lanes 0 .. 19 then store 16 + 4 contiguous bytes
  unsigned char* trw = wvb + 2 * SLOTB;
  const bool pl = (li & 1) == 0;                              // this lane holds a pooled pixel
  const uint32_t twA = keep_in_vgpr(lds_addr(trw + ((pl && lj < 3) ? ((li >> 1) * CH + 4 * lj) * 4 : 400)));
  const uint32_t twB = keep_in_vgpr(lds_addr(trw + ((pl && lj < 2) ? ((li >> 1) * CH + 4 * lj + 2) * 4 : 408)));
  const uint32_t tcA = keep_in_vgpr(lds_addr(trw + ((pl && lj < 3) ? 320 + (li >> 1) * CH + 4 * lj : 400)));
  const uint32_t tcB = keep_in_vgpr(lds_addr(trw + ((pl && lj < 2) ? 320 + (li >> 1) * CH + 4 * lj + 2 : 408)));
  const uint32_t trd = keep_in_vgpr(lds_addr(trw + (lane < 2 * CH ? 16 * lane : 0)));
  const uint32_t trc = keep_in_vgpr(lds_addr(trw + 320 + (lane < 2 * CH ? 4 * lane : 0)));
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)sb * a.out_bstride, 0, Hp * Wp * CH * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t amax_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_amax ? a.out_amax + (long)sb * Hp * Wp * CH : (uint8_t*)a.out, 0, a.out_amax ? Hp * Wp * CH : 0, 0x00020000);      // (forward-only networks: no codes)
  const unsigned eL = lane < 2 * CH ? (unsigned)(stile * 8 * CH * 4 + 16 * lane) : BIG;      // byte offset of the lane's four values within a pooled image row

  f32x4 acc[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) acc[s] = biasv;
  float tv[4] = {0.f, 0.f, 0.f, 0.f};                         // the even output row's x-pooled values and which pixel of the pair they came from
  bool xgt[4] = {false, false, false, false};

  x_store(0, 0);
  x_load(0, 2);
  read_x(0, 0);
  if (NCH > 1) read_x(NCH - 1, 0);
  __builtin_amdgcn_sched_barrier(0);

  auto step = [&](auto sqtag, auto gentag, const int q) __attribute__((always_inline)) {
    constexpr int SQ = decltype(sqtag)::value;
    constexpr bool GEN = decltype(gentag)::value;
    constexpr int SD = (SQ + 2 * NSET - P - 1) % NSET;       // the set of output row q - P - 1: complete since the last step
    constexpr int SLOT = SQ & 1;
    constexpr bool YODD = ((SQ + P + 1) & 1) != 0;           // output row q - P - 1 is odd: a pooled row is complete (q0 is a multiple of UNR: even)
    const int yd = q - P - 1;
    // ReLU + pool of output row yd (conv_rs16.h's rule: the first maximum in window order wins)
    auto pool = [&]() __attribute__((always_inline)) {
      float pv[4]; int code[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z0 = acc[SD][r], z1 = fwrs_swap1(z0);      // even lane: its pixel and the odd neighbour's
        const bool xgb = z1 > z0;
        const float bv = xgb ? z1 : z0;
        if (YODD) {
          const bool lower = bv > tv[r];
          const float mx = lower ? bv : tv[r];
          const bool act = mx > 0.f;
          code[r] = (lower ? (xgb ? 3 : 2) : (xgt[r] ? 1 : 0)) | (act ? POOL_ACTIVE : 0);
          pv[r] = act ? mx : 0.f;
        } else { tv[r] = bv; xgt[r] = xgb; }
      }
      if (YODD) {
        lds_store(twA, 0, (f32x2){pv[0], pv[1]});
        lds_store(twB, 0, (f32x2){pv[2], pv[3]});
        lds_store_u16(tcA, 0, (unsigned)(code[0] | (code[1] << 8)));
        lds_store_u16(tcB, 0, (unsigned)(code[2] | (code[3] << 8)));
      }
    };
    auto stores = [&]() __attribute__((always_inline)) {
      const f32x4 v = lds_load<f32x4>(trd, 0);
      const unsigned cd = lds_load<unsigned>(trc, 0);
      const bool live = !GEN || (yd >= 0 && yd < H);
      const unsigned eo = live ? eL : BIG;
      const int orow = (live ? (yd >> 1) : 0) * (Wp * CH);
      buffer_store_b128_held((k16_u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, out_rsrc, (int)eo, orow * 4);
      __builtin_amdgcn_raw_buffer_store_b32(cd, amax_rsrc, (int)(live ? (eL >> 2) : BIG), orow, 0);
    };
    auto mfmas = [&](auto chtag) __attribute__((always_inline)) {
      constexpr int ch = decltype(chtag)::value;
#pragma unroll
      for (int sum = ORDER; sum >= 0; --sum)                 // small products first; (pa, pb) = (2, ORDER - 2) is the first one issued
#pragma unroll
        for (int pa = 2; pa >= 0; --pa) {
          const int pb = sum - pa;
          if (pb >= 0 && pb <= 2) {
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
              const int s = (SQ + P - ky + NSET) % NSET;     // output row q + P - ky
              const bool restart = ch == 0 && sum == ORDER && pa == 2 && ky == 0;      // (the set output row q + P starts in: from the bias)
              acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k16_bf16x8, wv[ky][ch][pa]), __builtin_bit_cast(k16_bf16x8, xb[ch][pb]),
                                                               restart ? biasv : acc[s], 0, 0, 0);
            }
          }
        }
    };
    if (!GEN || q < H) {
      mfmas(std::integral_constant<int, 0>{});
      x_store((SQ + 1) & 1, SLOT ^ 1);                       // input row q + 1 (in the staging registers since two steps ago) -> LDS
      x_load((SQ + 1) & 1, q + 3);                           // (behind the last row: beyond the descriptor's range, zeros nobody uses)
      pool();
      read_x(0, SLOT ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (NCH > 1) {
        mfmas(std::integral_constant<int, 1>{});
        if (YODD) stores();
        read_x(1, SLOT ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      } else if (YODD) stores();
    } else { pool(); if (YODD) stores(); }                   // (steps behind the image: the last output rows)
  };
  auto block = [&](auto gentag, const int q0) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(gentag)::value;
    const int qe = H + P + 1;
    if (GEN && q0 + 0 >= qe) return; step(std::integral_constant<int, 0>{}, gentag, q0 + 0);
    if (GEN && q0 + 1 >= qe) return; step(std::integral_constant<int, 1>{}, gentag, q0 + 1);
    if (GEN && q0 + 2 >= qe) return; step(std::integral_constant<int, 2>{}, gentag, q0 + 2);
    if (GEN && q0 + 3 >= qe) return; step(std::integral_constant<int, 3>{}, gentag, q0 + 3);
    if constexpr (UNR == 6) {
      if (GEN && q0 + 4 >= qe) return; step(std::integral_constant<int, 4>{}, gentag, q0 + 4);
      if (GEN && q0 + 5 >= qe) return; step(std::integral_constant<int, 5>{}, gentag, q0 + 5);
    }
  };
  int q0 = 0;
  block(std::true_type{}, q0); q0 += UNR;
  for (; q0 + UNR <= H; q0 += UNR) block(std::false_type{}, q0);      // (every step multiplies a row and completes an output row of the image)
  for (; q0 < H + P + 1; q0 += UNR) block(std::true_type{}, q0);
}

template <int KSZ, int TPR, int ORDER>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_fw_rs_kernel(const ConvArgsN batch) {
  conv_fw_rs_body<KSZ, TPR, ORDER>(batch, blockIdx.x, blockIdx.y);
}

// conv3's (3x3) forward with bias + ReLU + 2x2 max-pool + arg-max code at rows of 32 / 64 pixels, 10 -> 10 channels
int conv_fw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, int epi, const ConvArgsN& a, bool* handled);

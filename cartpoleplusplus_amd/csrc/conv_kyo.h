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

// LDS accesses through explicit 32-bit addresses: a pointer into the dynamic LDS block is (relocated base + offset),
// and hipcc re-adds base and large constant offsets in VGPRs at every use.  f32 MFMAs do not overlap with VALU work
// on this chip, so the row loop keeps finished addresses in registers (made opaque once, outside the loop) and lets
// the ds_read / ds_write immediate offset field do the rest.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)p; }
__device__ __forceinline__ uint32_t keep_in_vgpr(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
template <typename T>
__device__ __forceinline__ T lds_load(uint32_t addr, int byte_off = 0) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T*>((uintptr_t)(addr + (uint32_t)byte_off));
}
template <typename T>
__device__ __forceinline__ void lds_store(uint32_t addr, int byte_off, const T& v) {
  *reinterpret_cast<__attribute__((address_space(3))) T*>((uintptr_t)(addr + (uint32_t)byte_off)) = v;
}
// A buffer store of MORE THAN 64 BITS reads its data registers some cycles after it issues.  LLVM keeps VALU writes of those registers
// two wait states away only when the store's soffset is NOT a register ("this hazard only exists if the instruction is not using a
// register in the soffset field", GCNHazardRecognizer) -- every row store of the row-streaming kernels has an SGPR soffset, and on gfx950
// a `v_max_f32 v202, |v202|, |v202|` one instruction behind `buffer_store_dwordx4 v[200:203], v174, s[0:3], s4 offen` changed what was
// stored, differently from run to run (round 6: conv_dx_rs.h at 64-wide rows with its row maximum live; profiles/NOTEBOOK_r06.md 10).  This
// store keeps its data registers alive and untouched for four more wait states; tools/check_store_data.py checks every listing for a
// wide store without that distance (`make check-stores`).
template <typename V4>
__device__ __forceinline__ void buffer_store_b128_held(const V4& d, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, voffset, soffset, 0);
  asm volatile("s_nop 3" :: "v"(d) : "memory");
}
// a 16-bit store into bytes that are READ BACK AS DWORDS: `unsigned short` and `unsigned` are different types to the compiler's type-based
// alias analysis, which lets the scheduler move the dword read above the halfword write (round 6: conv_rs16.h's pivots of an odd channel
// count were read before they were written, on the first rows only); a may_alias halfword aliases everything
typedef unsigned short __attribute__((may_alias)) lds_u16_alias;
__device__ __forceinline__ void lds_store_u16(uint32_t addr, int byte_off, unsigned v) {
  *reinterpret_cast<__attribute__((address_space(3))) lds_u16_alias*>((uintptr_t)(addr + (uint32_t)byte_off)) = (unsigned short)v;
}

// Arg-max code byte of a pooled cell: bits 0-1 = the window position dy * 2 + dx of the first maximum, bit 2 (POOL_ACTIVE) = the pooled
// output is > 0, i.e. the ReLU lets the cell's gradient through -- the backward kernels rebuild dY from (pooled gradient, code) and never
// read the pooled OUTPUT again (round 4; rounds 1-3 read it for this one bit: 21 MB per conv1-dW launch at cfg3).
constexpr int KYO_NO = 10;                         // filters per layer (base_network.py:103,111,119)

template <int CIN, int KS, int XT, int IPW>
struct KyoGeom {
  static constexpr int P = KS / 2;
  static constexpr int NT = (KS * KYO_NO + 15) / 16;        // N tiles over columns (p, o)
  static constexpr int STRIPS = 4 / IPW, SW = 16 * XT, WPAD = STRIPS * SW;
  static constexpr int KROW = KS * CIN;                     // k = (kx, c) of one input row
  // K order inside a row: lane group lj (= MFMA k index) owns the contiguous run [Q4*lj, Q4*lj + Q4) of the first
  // 4*Q4 values (Q4 even: its operand pairs are 8-byte aligned), step st < Q4 takes k = Q4*lj + st; the remaining
  // R < 8 values go to XS extra steps with k = 4*Q4 + 4*(st - Q4) + lj.  A lane's operands of consecutive steps are
  // contiguous in LDS (one ds_read2_b64 feeds 4 MFMAs) and every step's address is one register + an immediate.
  static constexpr int Q4 = (KROW / 4) & ~1, R = KROW - 4 * Q4, XS = (R + 3) / 4;
  static constexpr int KSTEPS = Q4 + XS;                    // = ceil(KROW / 4)
  static constexpr int NGT = (KSTEPS + 3) / 4;              // groups of up to 4 k-steps (one B fragment read each)
  static constexpr int FP = (4 - (P * CIN) % 4) % 4;        // front pad: image column 0 lands 16-byte aligned
  static constexpr int ROWF = ((FP + (WPAD + KS - 1) * CIN + 8) + 3) & ~3;   // floats per staged row (+ k over-read)
  static constexpr int WLF = KS * NGT * 4 * KYO_NO * 4;     // weight floats in LDS: [ky][group][lj][o][step]
  static constexpr int RING = 3;                            // staged rows in flight: row q is multiplied, q+1 is
                                                            // already visible (its first operands prefetch across the
                                                            // barrier), q+2 is being written
  static constexpr int EF = 2 * 8 * XT * KYO_NO * 2;        // per wave: (value, code) of the two rows of a pool pair
  static constexpr int WHF = 2 * ((CIN + 8 + 3) & ~3);      // whitening scale[] and shift[], each extended by 8 (wrap-around)
  static constexpr int LDS_FLOATS = WLF + RING * IPW * ROWF + 4 * EF + IPW * WHF;     // one whitening table per image
  static __host__ __device__ constexpr int steps(int g) { return KSTEPS - 4 * g < 4 ? KSTEPS - 4 * g : 4; }
  static __host__ __device__ constexpr int kmap(int st, int lj) {
    return st < Q4 ? Q4 * lj + st : 4 * Q4 + 4 * (st - Q4) + lj;
  }
};

// CHB: bytes per staging chunk -- 16 when the image rows are 16-byte multiples (64 x 64 x 18 f16), 8 or 4 otherwise (the
// reference's default 50 x 50 render: 1800-byte rows)
// PLAIN: write the plain conv output rows (no bias / ReLU / pool): batch norm needs the statistics of z first
template <int CIN, int KS, int XT, int IPW, int IN_MODE, int CHB = 16, bool PLAIN = false>
__device__ __forceinline__ void conv_fwd_kyo_body(const ConvArgsN& batch, const int bx, const int by) {
  typedef KyoGeom<CIN, KS, XT, IPW> G;
  typedef typename StageType<IN_MODE>::type ST;
  constexpr bool WHITEN = (IN_MODE == IN_F16_WHITEN || IN_MODE == IN_F32_WHITEN);
  // IN_DY: the dX pass of the layer -- input rows are dY rows rebuilt from the pooled gradient + arg-max code, the
  // weights are flipped and transposed, the epilogue writes plain full-resolution rows (no bias / ReLU / pool)
  constexpr bool DX = (IN_MODE == IN_DY);
  // IN_F32_FLIP: the same dX pass fed with DENSE dY rows (batch norm): plain f32 row staging, flipped weights, plain rows out
  constexpr bool FLIP = DX || (IN_MODE == IN_F32_FLIP);
  constexpr bool PLAIN_OUT = FLIP || PLAIN;
  constexpr int P = G::P, NT = G::NT, NGT = G::NGT, ROWF = G::ROWF, NO = KYO_NO;
  constexpr int EPC = CHB / (int)sizeof(ST);          // elements per staging chunk
  static_assert(EPC >= 1, "chunk smaller than an element");
  constexpr bool A64 = (CIN % 2 == 0) && (G::FP % 2 == 0);      // A operand pairs are 8-byte aligned in LDS (Q4 is even)
  constexpr int RING = G::RING, RSET = IPW * ROWF;              // row buffers in flight, floats per buffer
#ifdef KYO_CLOCK_PROBE
  const unsigned long long pe = __builtin_amdgcn_s_memrealtime();
#endif
  const ConvArgs& a = batch.a[by];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                                   // [KS][NGT][4][NO][4]
  float* rows = lds + G::WLF;                        // [RING][IPW][ROWF]
  float2* ebuf = reinterpret_cast<float2*>(rows + RING * RSET);   // [4 waves][2 parities][8*XT][NO] (value, code)
  float* whs = rows + RING * RSET + 4 * G::EF;       // per image: whitening scale[c mod CIN], c < CIN + 8; then shift[] likewise
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int img = wave / G::STRIPS, strip = wave % G::STRIPS;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;
  // launches that would leave most SIMDs with one wave (dX of two networks) split every image into bands of output rows
  // [ylo, yhi); a band also reads the P (even: 2) input rows above and below it
  const int nbands = a.nbands > 1 ? a.nbands : 1;
  const int band = bx % nbands;
  const int b0 = (bx / nbands) * IPW;
  const int ylo = nbands > 1 ? band * a.band_rows : 0, yhi = nbands > 1 ? min(H, ylo + a.band_rows) : H;
  const int qs = max(0, (ylo - P) & ~1), qe = min(H, yhi + P);      // input rows of the band

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
      const int k = G::kmap(4 * g + s, l);
      const bool ok = i < G::WLF && s < G::steps(g) && o < nout && k < G::KROW;
      // dX = correlation of dY with W'[ky][kx][c'][o'] = W[KS-1-ky][KS-1-kx][o'][c'], W stored (KS,KS,Cin = nout,Cout = CIN)
      const int widx = FLIP ? (((KS - 1 - ky) * KS + (KS - 1 - k / CIN)) * nout + o) * CIN + k % CIN : (ky * G::KROW + k) * nout + o;
      const float v = a.w[ok ? widx : 0];
      wv[n] = ok ? (a.wscale != 0.f ? v * a.wscale : v) : 0.f;
    }
#pragma unroll
    for (int n = 0; n < NW; ++n)
      if (tid + n * CONV_THREADS < G::WLF) wl[tid + n * CONV_THREADS] = wv[n];
  }

  // ---- row staging: chunk ch -> (image, 16-byte chunk of the row); a thread's chunks are the same for every row,
  // so the byte offset from the (uniform) row base, the LDS destination and the whitening constants' address are
  // fixed up front
  const int cpr = DX ? 1 : (W * CIN) / EPC;          // chunks per image row (W*CIN % EPC == 0 checked by the host)
  constexpr int NVMAX = (IPW * G::WPAD * CIN / EPC + CONV_THREADS - 1) / CONV_THREADS;
  uint4 sv[NVMAX];
  unsigned sbyte[NVMAX];                             // byte offset of the chunk from the row base of image b0
  bool sact[NVMAX];
  uint32_t sdst[NVMAX];                              // LDS address of the chunk in ring slot 0
  uint32_t swh[NVMAX];                               // LDS address of the whitening scale of the chunk's first channel
  if (WHITEN) {                                      // (white_bstride != 0: every image is whitened with its own statistics)
    for (int e = tid; e < IPW * (CIN + 8); e += CONV_THREADS) {
      const int im = e / (CIN + 8), c = e - im * (CIN + 8);
      const long wo = (b0 + im < a.B ? (long)(b0 + im) : 0) * a.white_bstride + c % CIN;
      whs[im * G::WHF + c] = a.scale[wo]; whs[im * G::WHF + G::WHF / 2 + c] = a.shift[wo];
    }
  }
#pragma unroll
  for (int i = 0; i < NVMAX; ++i) {
    const int ch = tid + CONV_THREADS * i;
    const int im = ch / cpr, j = ch - im * cpr;
    sact[i] = im < IPW && b0 + im < a.B;
    sbyte[i] = (unsigned)((long)im * a.in_bstride + j * EPC) * (unsigned)sizeof(ST);
    sdst[i] = keep_in_vgpr(lds_addr(rows + im * ROWF + G::FP + P * CIN + j * EPC));
    swh[i] = keep_in_vgpr(lds_addr(whs + (im < IPW ? im : 0) * G::WHF + (j * EPC) % CIN));
  }
  // buffer addressing: uniform descriptor (images b0 .. b0+IPW-1) + per-lane byte offset (constant) + scalar row
  // offset -- no vector address arithmetic per row
  const int nimg = a.B - b0 < IPW ? a.B - b0 : IPW;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      DX ? (void*)const_cast<float*>(a.dy.dpool) : (void*)const_cast<ST*>((const ST*)a.in + (long)b0 * a.in_bstride), 0,
      DX ? 0 : (int)(nimg * a.in_bstride * (long)sizeof(ST)), 0x00020000);
  const int rowbytes = W * CIN * (int)sizeof(ST);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // ---- IN_DY staging: a thread owns pooled cells (image, px, o) -- the same cells for every pooled row; one
  // (masked gradient, code) pair serves the two dY rows of its pooled row (requested one row ahead)
  constexpr int NCELL = DX ? (IPW * (G::WPAD / 2) * NO + CONV_THREADS - 1) / CONV_THREADS : 1;
  bool qact[NCELL]; uint32_t qdst[NCELL]; int qvo[NCELL], qvd[NCELL], qco[NCELL];
  float qg[NCELL], qdv[NCELL]; int qcode[NCELL], qrc[NCELL];
  const int dyWp = W >> 1, dyHp = H >> 1;
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_pool = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy.pool + (DX ? (long)b0 * a.dy.pool_bstride : 0)), 0, DX ? (int)(nimg * a.dy.pool_bstride * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dpool = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.dy.dpool + (DX ? (long)b0 * a.dy.dpool_bstride : 0)), 0, DX ? (int)(nimg * a.dy.dpool_bstride * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_amax = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a.dy.amax + (DX ? (long)b0 * dyHp * dyWp * CIN : 0)), 0, DX ? nimg * dyHp * dyWp * CIN : 0, 0x00020000);
  if (DX) {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      const int idx = tid + CONV_THREADS * c;
      const int per = dyWp * CIN;                     // cells of one pooled row of one image (dY has CIN channels)
      const int im = idx / per, e = idx - im * per;
      const int px = e / CIN, o = e - px * CIN;
      qact[c] = im < IPW && b0 + im < a.B;
      qdst[c] = keep_in_vgpr(lds_addr(rows + im * ROWF + G::FP + (2 * px + P) * CIN + o));
      qvo[c] = (int)((long)im * a.dy.pool_bstride + e) * 4;
      qvd[c] = (int)((long)im * a.dy.dpool_bstride + e) * 4;
      qco[c] = im * dyHp * dyWp * CIN + e;
      qg[c] = 0.f; qcode[c] = 0; qdv[c] = 0.f; qrc[c] = 0;
    }
  }
  auto dy_issue = [&](int py) {
    const bool rowok = py >= 0 && py < dyHp;         // uniform
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      qdv[c] = 0.f; qrc[c] = 0;
      if (rowok && qact[c]) {
        const int so = py * dyWp * CIN;
        qdv[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_dpool, qvd[c], so * 4, 0));
        qrc[c] = __builtin_amdgcn_raw_buffer_load_b8(rs_amax, qco[c], so, 0);
      }
    }
  };
  auto dy_conv = [&]() {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) { qg[c] = (qrc[c] & POOL_ACTIVE) ? qdv[c] : 0.f; qcode[c] = qrc[c] & 3; }
  };
  auto dy_store = [&](int slot, int ry) {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      if (qact[c]) {
        const float v0 = qcode[c] == 2 * ry ? qg[c] : 0.f, v1 = qcode[c] == 2 * ry + 1 ? qg[c] : 0.f;
        lds_store(qdst[c], slot * RSET * 4, v0);
        lds_store(qdst[c], slot * RSET * 4 + CIN * 4, v1);
      }
    }
  };
  // dY row y into ring slot `slot`: even rows convert the cells requested one row earlier, odd rows request the next
  auto dx_stage_row = [&](int y, int slot) {
    if ((y & 1) == 0) dy_conv();
    dy_store(slot, y & 1);
    if ((y & 1) == 1) dy_issue((y + 1) >> 1);
  };
  auto stage_load = [&](int y) {
    if (DX) return;
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      if (sact[i]) {
        if (CHB == 16) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, (int)sbyte[i], y * rowbytes, 0);
          sv[i] = make_uint4(v.x, v.y, v.z, v.w);
        } else if (CHB == 8) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)sbyte[i], y * rowbytes, 0);
          sv[i] = make_uint4(v.x, v.y, 0u, 0u);
        } else {
          sv[i] = make_uint4(__builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)sbyte[i], y * rowbytes, 0), 0u, 0u, 0u);
        }
      }
    }
  };
  auto stage_store = [&](int slot, int y) {          // slot: uniform ring slot; y: the row being written
    if (DX) { dx_stage_row(y, slot); return; }
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      if (sact[i]) {
        float x[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) x[k] = ChunkOps<ST>::get(sv[i], k);
        if (WHITEN) {
#pragma unroll
          for (int k = 0; k < EPC; k += 2) {         // pairs are 8-byte aligned when CIN is even
            if (k + 1 < EPC) {
              f32x2 sc, sh;
              if (CIN % 2 == 0) {
                sc = lds_load<f32x2>(swh[i], 4 * k);
                sh = lds_load<f32x2>(swh[i], 4 * (G::WHF / 2 + k));
              } else {
                sc = (f32x2){lds_load<float>(swh[i], 4 * k), lds_load<float>(swh[i], 4 * k + 4)};
                sh = (f32x2){lds_load<float>(swh[i], 4 * (G::WHF / 2 + k)), lds_load<float>(swh[i], 4 * (G::WHF / 2 + k) + 4)};
              }
              x[k] = x[k] * sc.x + sh.x;
              x[k + 1] = x[k + 1] * sc.y + sh.y;
            } else {
              x[k] = x[k] * lds_load<float>(swh[i], 4 * k) + lds_load<float>(swh[i], 4 * (G::WHF / 2 + k));
            }
          }
        }
        const uint32_t dst = sdst[i] + (uint32_t)(slot * RSET * 4);
        if (EPC >= 4) {
#pragma unroll
          for (int k = 0; k + 3 < EPC; k += 4) lds_store(dst, 4 * k, (f32x4){x[k], x[k + 1], x[k + 2], x[k + 3]});
        } else if (EPC == 2) {
          lds_store(dst, 0, (f32x2){x[0], x[1]});
        } else {
          lds_store(dst, 0, x[0]);
        }
      }
    }
  };

  // ---- per-lane column bookkeeping: column j = 16 t + li = p * NO + o of N tile t.  The row loop is unrolled by KS,
  // so ky of a block in unrolled step sq is (sq + P - p) mod KS for every pass of the loop: one LDS address per
  // (step, tile) held in registers replaces all per-row address arithmetic for the rotating weights.
  uint32_t wadr[KS][NT];
  uint32_t eadr[NT];                                 // pool-pair buffer slot of (pooled x = 2 lj, o), parity 0
  float biast[NT];
  const int swave = __builtin_amdgcn_readfirstlane(wave);       // scalar copies: uniform address arithmetic below
  const int simg = swave / G::STRIPS, sstrip = swave % G::STRIPS;
  const int sbimg = b0 + simg;
  float2* ev = ebuf + swave * (2 * 8 * XT * NO);     // [2 parities][8*XT][NO]
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = 16 * t + li;
    const bool valid = j < KS * NO;
    const int o = j % NO;
    biast[t] = (!PLAIN_OUT && valid && o < nout) ? a.bias[o] : 0.f;
    eadr[t] = PLAIN_OUT ? (uint32_t)(((sstrip * G::SW + 4 * lj) * nout + o) * 4)      // byte offset of (x = strip + 4 lj, o) in an output row
                 : keep_in_vgpr(lds_addr(ev + (lj * 2) * NO + o));
    const int p = valid ? j / NO : 0;
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int ky = (sq + P - p + KS) % KS;
      wadr[sq][t] = keep_in_vgpr(lds_addr(wl + ((ky * NGT * 4 + lj) * NO + (valid ? o : 0)) * 4));   // (ky, group 0, lj, o, step 0)
    }
  }
  constexpr int GBYTES = 4 * NO * 4 * 4;             // bytes between the groups of one ky plane

  // ---- pooled-row writer: entry idx = xl * nout + o of this wave's 8*XT pooled columns, two entries per lane at most
  constexpr int NC = (8 * XT * NO + 63) / 64;
  uint32_t cadr[NC];
  bool cact[NC];
  unsigned coe[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = lane + 64 * i;
    const int xl = idx / nout, o = idx - xl * nout;
    const int px = ((sstrip * G::SW) >> 1) + xl;
    cact[i] = idx < 8 * XT * nout && px < Wp && sbimg < a.B;
    cadr[i] = keep_in_vgpr(lds_addr(ev + (cact[i] ? xl * NO + o : 0)));
    coe[i] = (unsigned)(px * nout + o);
  }
  const int simg_ok = sbimg < a.B ? sbimg : 0;
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out + (long)simg_ok * a.out_bstride, 0, (PLAIN_OUT ? H * W : Hp * Wp) * nout * 4, 0x00020000);   // uniform (per wave)
  const __amdgpu_buffer_rsrc_t amax_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_amax + (long)simg_ok * Hp * Wp * nout, 0, Hp * Wp * nout, 0x00020000);

  // accumulators start at (and are reset to) the bias of their column: conv + bias without an add in the epilogue
  f32x4 acc[XT][NT];
#pragma unroll
  for (int m = 0; m < XT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};

  __syncthreads();                                   // zeroed row buffers + whitening table visible
  if (DX) dy_issue(qs >> 1);
  for (int r = 0; r < RING - 1; ++r) {
    if (qs + r < qe) { stage_load(qs + r); stage_store(r, qs + r); }
  }
  if (qs + RING - 1 < qe) stage_load(qs + RING - 1);
  __syncthreads();

  int rcur = 0;                                      // ring slot of row q (uniform)

  // operands of one group of k-steps; two sets so that the loads of group g+1 are in flight under the MFMAs of g
  float av[2][XT][4];
  f32x4 bv[2][NT];
  auto load_ops = [&](int g, int set, uint32_t ab, uint32_t ax, const uint32_t* wa) {
#pragma unroll
    for (int m = 0; m < XT; ++m) {
      const int mo = 4 * (m * 16 * CIN);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int st = 4 * g + s;
#ifdef KYO_ABL_NOLDSA
        av[set][m][s] = biast[0];
#else
        if (s >= G::steps(g)) {
          av[set][m][s] = 0.f;
        } else if (st >= G::Q4) {
          av[set][m][s] = lds_load<float>(ax, mo + 4 * (4 * (st - G::Q4)));
        } else if (A64 && (s & 1) == 0 && st + 1 < G::Q4) {
          const f32x2 u = lds_load<f32x2>(ab, mo + 4 * st);
          av[set][m][s] = u.x; av[set][m][s + 1] = u.y;
        } else if (!(A64 && (s & 1) == 1 && st < G::Q4)) {
          av[set][m][s] = lds_load<float>(ab, mo + 4 * st);
        }
#endif
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#ifdef KYO_ABL_NOLDSB
      bv[set][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};
#else
      bv[set][t] = lds_load<f32x4>(wa[t], g * GBYTES);
#endif
    }
  };
  // operand k = (kx, c) of output column x starts at x * CIN of the staged row; the lane-group part of k is folded
  // into the two address registers (main run, extra steps)
  const float* const arow0 = rows + img * ROWF + G::FP + (strip * G::SW + li) * CIN;
  const uint32_t arow = keep_in_vgpr(lds_addr(arow0 + G::Q4 * lj));
  const uint32_t axtr = keep_in_vgpr(lds_addr(arow0 + 4 * G::Q4 + lj));
#ifdef KYO_CLOCK_PROBE
  const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
  // The row loop is unrolled by KS so that the block that completes in a step -- and with it the (tile, lane range)
  // that is taken out and reset -- is a compile-time constant: no per-lane block compares or select chains.
  for (int q0 = (qs / KS) * KS; q0 < yhi + P; q0 += KS) {
    {  // issue arbitration is by priority, then age: without this the oldest of the co-resident workgroups (one per
       // network) runs ahead and the youngest finishes alone; rotating the priority every KS rows keeps them level
       // (0.3786 -> 0.3745 ms for conv1)
      const int pr = (by + q0 / KS) & 3;
      if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
      else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int q = q0 + sq;
      if (q < qs) continue;                          // uniform: rows above the band
      if (q >= yhi + P) break;                       // uniform
      constexpr int PD_BASE = (KS - P) % KS;
      const int pdone = (PD_BASE + sq) % KS;         // block of row y = q - P: constant after unrolling
      const int rnext = rcur + 1 == RING ? 0 : rcur + 1;
#ifndef KYO_ABL_NOSTAGE
      {                                              // row q + RING - 1 -> the slot row q - 1 has just left
        const int rw = rcur == 0 ? RING - 1 : rcur - 1;
        if (q + RING - 1 < qe) stage_store(rw, q + RING - 1);
        if (q + RING < qe) stage_load(q + RING);
      }
#endif

      if (q < qe) {
        const uint32_t ab = arow + (uint32_t)(rcur * RSET * 4), ax = axtr + (uint32_t)(rcur * RSET * 4);
        if (q == qs) load_ops(0, 0, ab, ax, wadr[sq]);   // (every later row: prefetched under the previous epilogue)
#pragma unroll
        for (int g = 0; g < NGT; ++g) {
          if (g + 1 < NGT) load_ops(g + 1, (g + 1) & 1, ab, ax, wadr[sq]);
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
      }
      // group 0 of the next row (made visible by the previous barrier) loads under the epilogue
      if (q + 1 < qe) {
        load_ops(0, NGT & 1, arow + (uint32_t)(rnext * RSET * 4), axtr + (uint32_t)(rnext * RSET * 4), wadr[(sq + 1) % KS]);
        if (NGT & 1) {
#pragma unroll
          for (int m = 0; m < XT; ++m)
#pragma unroll
            for (int s = 0; s < 4; ++s) av[0][m][s] = av[1][m][s];
#pragma unroll
          for (int t = 0; t < NT; ++t) bv[0][t] = bv[1][t];
        }
      }

      // ---- output row y = q - P is complete (rows y < 0 do not exist: their block is only reset): the lanes of block
      // pdone in the one or two tiles it spans do the x half of the max-pool and park (value, code) in the pool-pair
      // buffer; the block restarts from the bias
#ifdef KYO_ABL_NOEPI
      const int y = (q == yhi + P - 1) ? q - P : -1;
#else
      const int y = q - P;
#endif
      const int par = y & 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int clo = pdone * NO - 16 * t, chi = pdone * NO + NO - 1 - 16 * t;    // block columns relative to tile t
        if (chi >= 0 && clo <= 15) {                 // compile-time: the block has columns in this tile
          const bool inr = li >= clo && li <= chi;
          if (inr) {
#pragma unroll
            for (int m = 0; m < XT; ++m) {
              if (PLAIN_OUT) {
                if (y >= ylo && sbimg < a.B) {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (FLIP || sstrip * G::SW + m * 16 + 4 * lj + r < W)      // dX rows tile the strips exactly (dispatch)
                      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][t][r]), out_rsrc,
                                                            (int)eadr[t] + ((m * 16 + r) * NO) * 4, y * W * nout * 4, 0);
                }
              } else if (y >= ylo) {
                const uint32_t ea = eadr[t] + (uint32_t)(par * (8 * XT * NO) * 8);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const float z0 = acc[m][t][2 * h], z1 = acc[m][t][2 * h + 1];
                  lds_store(ea, ((m * 8 + h) * NO) * 8, (f32x2){z1 > z0 ? z1 : z0, __int_as_float(z1 > z0 ? 1 : 0)});
                }
              }
              acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};
            }
          }
        }
      }
      if (!PLAIN_OUT && y >= ylo && par == 1 && (y >> 1) < Hp) {   // wave-uniform: both rows of a pool pair are in the buffer
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int orow = (y >> 1) * Wp * nout;         // uniform
#pragma unroll
        for (int i = 0; i < NC; ++i) {
          if (cact[i]) {
            const f32x2 top = lds_load<f32x2>(cadr[i], 0), bot = lds_load<f32x2>(cadr[i], (8 * XT * NO) * 8);
            const bool lower = bot.x > top.x;
            const float mx = lower ? bot.x : top.x;
            const int code = lower ? 2 + __float_as_int(bot.y) : __float_as_int(top.y);
#ifdef KYO_ABL_NOSTORE
            if (mx == 123.456f)
#endif
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mx > 0.f ? mx : 0.f), out_rsrc, (int)(coe[i] * 4), orow * 4, 0);
#if defined(KYO_ABL_NOSTORE) || defined(KYO_ABL_NOCODE)
            if (mx == 123.456f)
#endif
            __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(code | (mx > 0.f ? POOL_ACTIVE : 0)), amax_rsrc, (int)coe[i], orow, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      rcur = rnext;
#ifndef KYO_ABL_NOBAR
      __syncthreads();
#endif
    }
  }
#ifdef KYO_CLOCK_PROBE
  if (tid == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) {
    const unsigned long long pc1 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
    printf("KYOCLK block %d,%d: %llu core cycles in loop; ref ticks: entry %llu loop %llu end %llu\n", (int)blockIdx.x, (int)blockIdx.y,
           pc1 - pc0, pe, pr0, pr1);
  }
#endif
}

template <int CIN, int KS, int XT, int IPW, int IN_MODE, int CHB, bool PLAIN>
__global__ __launch_bounds__(CONV_THREADS, (XT == 1 ? 4 : 2)) void conv_fwd_kyo_kernel(const ConvArgsN batch) {
  conv_fwd_kyo_body<CIN, KS, XT, IPW, IN_MODE, CHB, PLAIN>(batch, blockIdx.x, blockIdx.y);
}

template <int CIN, int KS, int XT, int IPW, int IN_MODE, int CHB = 16, bool PLAIN = false>
static inline int conv_fwd_kyo_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef KyoGeom<CIN, KS, XT, IPW> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
  auto kern = conv_fwd_kyo_kernel<CIN, KS, XT, IPW, IN_MODE, CHB, PLAIN>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int grid = ((a.B + IPW - 1) / IPW) * (a.nbands > 1 ? a.nbands : 1);
  if (ctx->pair && CIN == 10 && IN_MODE == IN_DY && CHB == 16 && !PLAIN && XT == 1 &&
      ((ctx->pair->layer == 2 && KS == 3 && IPW == 4) || (ctx->pair->layer == 1 && KS == 5 && IPW == 2))) {      // leaves with the layer's dW (conv*_bwd_pair.hip)
    ctx->pair->dx = batch; ctx->pair->dx_gx = grid; ctx->pair->dx_lds = lds_bytes; ctx->pair->have_dx = true;
    return 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

int conv_fwd_kyo_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled);
int conv_fwd_kyo_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled);
int conv_fwd_kyo_dispatch_plain(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, bool* handled);

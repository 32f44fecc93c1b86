// conv2's and conv3's weight / bias gradient on the bf16 matrix pipes, ONE WAVE PER UNIT ("dwrs"): v_mfma_f32_16x16x32_bf16.
//
//   dW[ky][kx][c][o] = sum_{b,q,x} in[b, q, x + kx - P, c] * dZ[b, q - ky + P, x, o]          (base_network.py:111-123's convs, backward)
//
// conv_dwb16.h's arithmetic (both f32 operands as three bf16 pieces, the six -- CPP_PRECISION_EXACT: nine -- largest piece products,
// f32 accumulation) and its operand layouts (the input row at a pixel pitch of CP halves, read through ds_read_b64_tr_b16; dZ rows as
// [piece][o][lane group][8 halves] in the same pixel order), in a different division of labour.  There the four waves of a workgroup
// share one (image, band) unit, each owning one 16-column tile of (ky, o): 24 MFMAs per wave and input row between two workgroup
// barriers, every A fragment read by all four waves, the staging of the next rows in front of the MFMAs.  Here
//
//   * a WAVE owns a unit and all 4 x 4 (3x3: 3 x 2) accumulator tiles of D[m = (kx, c)][n = (ky, o)] (64 VGPRs): 96 (36) MFMAs per input row, every
//     A fragment read once, no barrier in the row loop -- the rows of a unit are staged in the wave's own LDS slots;
//   * the staging work (pooled gradient + arg-max code -> dZ row -> three bf16 planes; f32 activations -> three planes) is dealt out
//     between the MFMAs (one wave per SIMD beside conv_dx_rs.h's: nobody else would issue in its place);
//   * the bias gradient is row (kx = P, c = CIN) x column (ky = P, o) of the same product: a "ones" channel in the input row's first plane;
//   * the four waves of a workgroup add their accumulators through LDS in a fixed order: one partial per workgroup (a quarter of
//     conv_dwb16.h's partials for conv_dw_reduce_kernel).
// Rows of 32 or 64 pixels (a unit = a 32-pixel column of a band: the contraction's one chunk per row; 3x3 also 16-wide rows, half a chunk),
// 10 -> 10 channels.
#pragma once
#include <type_traits>
#include "conv_dwb16.h"

template <int KS_>
struct DwRsGeom {
  static constexpr int KS = KS_, P = KS / 2, PB = 2, CIN = KYO_NO, NO = KYO_NO, WC = 32;      // WC: a unit's column of an image, 32 pixels; PB: P rounded up to even
  static constexpr int CP = 12;                               // channel pitch of a pixel in LDS (halves): 10 channels, the ones channel, one spare
  static constexpr int MT = (KS * CP + 15) / 16, NT = (KS * NO + 15) / 16;      // 16-row tiles of m = CP kx + c, 16-column tiles of n = NO ky + o (5x5: 4 x 4; 3x3: 3 x 2)
  static constexpr int ROWB = 2 * (((CP * (WC + KS - 1) + 16 * MT - KS * CP) + 7) & ~7);      // a plane of a staged input row: (WC + 2 P) pixels x CP halves + the m over-read, bytes
  static constexpr int XSLOT = 3 * ROWB;
  static constexpr int NXS = 2;                               // input rows in LDS: the one being multiplied, the one being written
  static constexpr int DOST = 80, DPC = NO * DOST;             // dZ row: [piece][o][4 lane groups x 16 bytes + skew]
  // (slots padded to 224 mod 256 bytes, which spreads a B read's 16 columns -- two or three ky, i.e. ring slots -- over 16 different bank
  // groups: no change, 3159 / 3160 / 3178 vs 3159 / 3165 / 3150 steps/s alternating on one box; the flat layout stays)
  static constexpr int DSLOT = 3 * DPC;
  static constexpr int NDS = KS + 1;                          // dZ rows in LDS: 2 P + 1 in use, one being written
  static constexpr int WVB = NXS * XSLOT + NDS * DSLOT;       // per wave
  static constexpr int LDS_BYTES = 4 * WVB;
  static constexpr int NW = KS * KS * CIN * NO;
  static constexpr int UNR = KS == 5 ? 6 : 12;                // steps per unrolled block: a multiple of NDS (slots), 2 (input slots) and 6 (three pooled rows in flight)
  static_assert(KS == 5 || KS == 3, "5x5 (conv2) or 3x3 (conv3)");
  static_assert(ROWB % 16 == 0 && WVB % 16 == 0 && WVB >= MT * NT * 4 * 64 * 4, "the wave's slots also hold its accumulators for the final sum");
};

// units = (image, band of `band` input rows, 32-pixel column); unit u of the launch's network `by` is wave (u % 4) of workgroup u / 4;
// units_per_img = bands x columns, column fastest.
template <int KSZ, int ORDER>
__device__ __forceinline__ void conv_dw_rs_body(const ConvArgsN& batch, const int units_per_img, const int band, const int bx, const int by) {
  typedef DwRsGeom<KSZ> G;
  constexpr int KS = G::KS, P = G::P, PB = G::PB, CIN = G::CIN, NO = G::NO, WC = G::WC, CP = G::CP, MT = G::MT, NT = G::NT;
  constexpr int ROWB = G::ROWB, XSLOT = G::XSLOT, DOST = G::DOST, DPC = G::DPC, DSLOT = G::DSLOT, NDS = G::NDS;
  constexpr unsigned BIG = 0x08000000u;
  const ConvArgs& a = batch.a[by];
  extern __shared__ __attribute__((aligned(16))) unsigned char dwrs_lds[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lj = lane >> 4;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* wvb = dwrs_lds + swave * G::WVB;
  unsigned char* xring = wvb;                                 // [NXS][3 planes][ROWB]
  unsigned char* dzring = wvb + G::NXS * XSLOT;               // [NDS][3 pieces][NO][DOST]
  const int H = a.H, Hp = H >> 1, W = a.W, Wp = W >> 1, ncol = (W + WC - 1) / WC;
  const int units = a.B * units_per_img;
  const int unit = bx * 4 + swave;
  const bool work = unit < units;
  const int ub = work ? unit / units_per_img : 0;
  const int uin = unit - ub * units_per_img;
  const int x0 = work ? (uin % ncol) * WC : 0;               // the unit's pixels: x0 .. x0 + 31
  const int q_lo = work ? (uin / ncol) * band : 0;
  const int rows = work ? min(band, H - q_lo) : 0;            // (band and q_lo are even)
  const int y0 = q_lo - PB;                                   // dZ row of ring position 0 (even: a pooled row's first)

  // ---- the wave's slots: zero; the ones channel (first plane, channel CIN = bf16 1.0) of the in-image pixels of both input slots
  for (int i = lane; i < G::WVB / 16; i += 64) reinterpret_cast<k16_u32x4*>(wvb)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
  for (int i = lane; i < G::NXS * WC; i += 64) {
    const int s = i / WC, x = i - s * WC;
    *reinterpret_cast<unsigned short*>(xring + s * XSLOT + 2 * (CP * (x + P) + CIN)) = (unsigned short)0x3F80u;
  }

  // ---- input rows: the unit's pixels and P more on each side (the neighbouring column's, or the SAME padding's zeros -- out of the
  // descriptor's range); lane owns channel pairs (wx, 2 cp) of that window, idx = lane + 64 i = 5 wx + cp: 8 contiguous bytes each
  constexpr int NXV = 3;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>((const float*)a.in + (long)ub * a.in_bstride), 0, work ? H * W * CIN * 4 : 0, 0x00020000);
  unsigned xvo[NXV]; uint32_t xdst[NXV];
#pragma unroll
  for (int i = 0; i < NXV; ++i) {
    const int idx = lane + 64 * i, wx = idx / 5, cp = idx - 5 * wx;
    const int x = x0 - P + wx;
    const bool on = idx < (WC + 2 * P) * 5;
    xvo[i] = (on && x >= 0 && x < W) ? (unsigned)(8 * (5 * x + cp)) : BIG;
    xdst[i] = keep_in_vgpr(lds_addr(xring + (on ? 2 * (CP * wx + 2 * cp) : ROWB - 8)));      // (idle lanes: zeros into the plane's tail)
  }
  f32x2 xraw[2][NXV];                                         // two rows in flight
  auto x_load = [&](const int buf, const int q) __attribute__((always_inline)) {      // (the row offset in the VGPR: the range check does not see soffset)
    const unsigned ro = q < H ? (unsigned)(q * (W * CIN * 4)) : BIG;      // (a scalar select: behind the image every lane is out of range)
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      const dw16_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (int)(xvo[i] + ro), 0, 0);
      xraw[buf][i] = (f32x2){__uint_as_float(v.x), __uint_as_float(v.y)};
    }
  };
  auto x_store1 = [&](const int buf, const int slot, const int i) __attribute__((always_inline)) {
    {
      unsigned short h0, m0, l0, h1, m1, l1;
      dwb_split3(xraw[buf][i][0], h0, m0, l0); dwb_split3(xraw[buf][i][1], h1, m1, l1);
      lds_store(xdst[i], slot * XSLOT, (unsigned)h0 | ((unsigned)h1 << 16));
      lds_store(xdst[i], slot * XSLOT + ROWB, (unsigned)m0 | ((unsigned)m1 << 16));
      lds_store(xdst[i], slot * XSLOT + 2 * ROWB, (unsigned)l0 | ((unsigned)l1 << 16));
    }
  };
  auto x_store = [&](const int buf, const int slot) __attribute__((always_inline)) { x_store1(buf, slot, 0); x_store1(buf, slot, 1); x_store1(buf, slot, 2); };

  // ---- dZ rows: lane l < 40 owns channel o = l % 10 of lane group g = l / 10: the 16 bytes one lane of the B operand reads -- pixels
  // 16 (g & 1) + 2 (g >> 1) + 4 j + r, i.e. both pixels of the pooled cells px_j = 8 (g & 1) + (g >> 1) + 2 j, j = 0 .. 3
  const int zg = lane / NO, zo = lane - zg * NO;
  const bool zon = lane < 4 * NO && x0 / 2 + 8 * (zg & 1) + (zg >> 1) + 6 < Wp;      // (16-wide rows: the second half of the chunk is padding)
  const __amdgpu_buffer_rsrc_t dp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy.dpool + (long)ub * a.dy.dpool_bstride), 0,
                                                                           work ? Hp * Wp * NO * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t am_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dy.amax + (long)ub * Hp * Wp * NO), 0,
                                                                           work ? Hp * Wp * NO : 0, 0x00020000);
  const unsigned zoff = zon ? (unsigned)((x0 / 2 + 8 * (zg & 1) + (zg >> 1)) * NO + zo) : BIG;      // element offset of cell j = 0 in a pooled row; cell j: + 2 j NO
  const uint32_t zdst = keep_in_vgpr(lds_addr(dzring + (zon ? zo * DOST + zg * 16 : DOST - 16)));      // (idle lanes: zeros into the skew)
  float zrg[3][4]; unsigned zrc[3][4];                        // three pooled rows in flight (conv_dx_rs.h)
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) { zrg[b][j] = 0.f; zrc[b][j] = 0u; }
  unsigned zp01[3] = {0u, 0u, 0u}, zp23[3] = {0u, 0u, 0u};    // the pieces of cells (0, 1) and (2, 3), packed
  unsigned zm01[4] = {0u, 0u, 0u, 0u}, zm23[4] = {0u, 0u, 0u, 0u};      // per window position 2 ry + rx: which halves belong to that pixel
  auto z_load = [&](const int buf, const int py) __attribute__((always_inline)) {
    const unsigned ro = (py >= 0 && py < Hp) ? (unsigned)(py * (Wp * NO)) : BIG;      // (a scalar select, no branch around the loads)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned e = zoff + ro + (unsigned)(2 * j * NO);
      zrg[buf][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dp_rsrc, (int)(e * 4u), 0, 0));
      zrc[buf][j] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(am_rsrc, (int)e, 0, 0);
    }
  };
  auto z_convert_half = [&](const int buf, const int hf) __attribute__((always_inline)) {      // cells (0, 1) or (2, 3) of the lane's four
    unsigned short pc[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float g = (zrc[buf][2 * hf + j] & POOL_ACTIVE) ? zrg[buf][2 * hf + j] : 0.f;
      dwb_split3(g, pc[j][0], pc[j][1], pc[j][2]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned v = (unsigned)pc[0][p] | ((unsigned)pc[1][p] << 16);
      if (hf == 0) zp01[p] = v; else zp23[p] = v;
    }
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) {
      const unsigned v = ((zrc[buf][2 * hf] & 3u) == (unsigned)pos ? 0xFFFFu : 0u) | ((zrc[buf][2 * hf + 1] & 3u) == (unsigned)pos ? 0xFFFF0000u : 0u);
      if (hf == 0) zm01[pos] = v; else zm23[pos] = v;
    }
  };
  auto z_convert = [&](const int buf) __attribute__((always_inline)) { z_convert_half(buf, 0); z_convert_half(buf, 1); };
  auto z_store = [&](const int slot, const int ry) __attribute__((always_inline)) {      // dZ row 2 py + ry of the converted pooled row
#pragma unroll
    for (int p = 0; p < 3; ++p)
      lds_store(zdst, slot * DSLOT + p * DPC, (k16_u32x4){zp01[p] & zm01[2 * ry], zp23[p] & zm23[2 * ry], zp01[p] & zm01[2 * ry + 1], zp23[p] & zm23[2 * ry + 1]});
  };

  // ---- MFMA operands (pixel dealing and transpose reads as conv_dw16.h / conv_dwb16.h)
  const int tj = (lane >> 2) & 3, tq = lane & 3;
  const uint32_t aadr = keep_in_vgpr(lds_addr(xring + 2 * (CP * (16 * (lj & 1) + 4 * tj + 2 * (lj >> 1)) + 4 * tq)));
  // column n = 16 nt + li = NO ky + o reads dZ ring position t - ky + P + PB at input row t of the band: slot (sq - ky + P + PB) mod NDS
  uint32_t bbase[NT]; int bky[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = 16 * nt + li;
    const bool nvalid = n < KS * NO;
    bky[nt] = nvalid ? n / NO : 0;
    bbase[nt] = lds_addr(dzring + (nvalid ? n % NO : 0) * DOST + lj * 16);
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (work) {
    // ---- prologue: dZ ring positions 0 .. PB + P (rows y0 ..: pooled rows y0 / 2 ..), input row q_lo; two more of each in flight
    const int py0 = y0 >> 1;                                  // (y0 is even; -1 for the first band: zeros)
    z_load(0, py0); z_load(1, py0 + 1); z_load(2, py0 + 2);
    x_load(0, q_lo); x_load(1, q_lo + 1);
#pragma unroll
    for (int d = 0; d <= PB + P; ++d) {                       // (the position behind the last one is stored by the first step)
      if ((d & 1) == 0) z_convert((d / 2) % 3);
      z_store(d % NDS, d & 1);
    }
    z_load(0, py0 + 3); z_load(1, py0 + 4);
    x_store(0, 0);
    x_load(0, q_lo + 2);
    __builtin_amdgcn_sched_barrier(0);

    // Step t (sq = t mod 6, compile time): multiply input row q_lo + t (slot t & 1) with dZ positions t .. t + 2 P; meanwhile dZ position
    // t + 2 P + 1 (row y0 + t + 5: pooled row py0 + (t + 5) / 2, buffer ((t + 5) / 2) mod 3) and input row q_lo + t + 1 go to LDS, and
    // the loads of pooled row py0 + (t + 5) / 2 + 3 (every second step) and of input row q_lo + t + 3 leave.
    // The row's 96 MFMAs run as 8 BLOCKS of 12: column tiles (0, 1), then (2, 3), each against the four row tiles.  Every operand is
    // requested a block (A: 6 transpose reads) or half a row (B: the other half's 6 reads, into the registers that half has just
    // released) before its use, and the staging work is dealt out over the blocks, between their MFMAs -- this wave is alone on its
    // SIMD slot: what it does not issue in an MFMA's shadow leaves the matrix pipe idle.
    k16_u32x4 bq[NT][3];
    k16_u32x4 av[3];
    auto b_load = [&](const int sq, const int nt) __attribute__((always_inline)) {      // column tile nt's operands for the step with t mod 6 = sq
      int sl = (sq + PB + P) % NDS - bky[nt]; sl = sl < 0 ? sl + NDS : sl;               // slot of this lane's column: ky differs per lane
      const uint32_t ad = bbase[nt] + (uint32_t)(sl * DSLOT);
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[nt][p] = lds_load<k16_u32x4>(ad, p * DPC);
    };
    auto a_load = [&](k16_u32x4 (&dst)[3], const int xs, const int mt) __attribute__((always_inline)) {
#pragma unroll
      for (int pa = 0; pa < 3; ++pa) {
        const int off = xs * XSLOT + pa * ROWB + mt * 32;
        const dw16_v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)off)));
        const dw16_v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)(off + 2 * CP))));
        const dw16_u32x2 u0 = __builtin_bit_cast(dw16_u32x2, r0), u1 = __builtin_bit_cast(dw16_u32x2, r1);
        dst[pa] = (k16_u32x4){u0.x, u0.y, u1.x, u1.y};
      }
    };
    if constexpr (KS == 5) { b_load(0, 0); b_load(0, 1); a_load(av, 0, 0); }
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](auto sqtag, const int t) __attribute__((always_inline)) {
      constexpr int SQ = decltype(sqtag)::value;
      constexpr int XS = SQ & 1;
      constexpr int ZPOS = SQ + PB + P + 1;                   // ring position being written (mod: 6 rows = three pooled rows)
      constexpr bool ZODD = (ZPOS & 1) != 0;                  // its image-row parity ry (y0 is even)
      constexpr int ZBUF = (ZPOS / 2) % 3;
      constexpr int NPROD = ORDER == B16_NINE ? 9 : 6;
      if constexpr (KS == 3) {
        // 3x3 (conv3): 36 MFMAs a row against the same staging work -- the row is bound by that work whatever the order; plain sequence
        // (bq / av requested at the top: the previous step left nothing in flight)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_load(SQ, nt);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          a_load(av, XS, mt);
#pragma unroll
          for (int sum = ORDER; sum >= 0; --sum)
#pragma unroll
            for (int pa = 2; pa >= 0; --pa) {
              const int pb = sum - pa;
              if (pb >= 0 && pb <= 2) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                  acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dwb_bf16x8, av[pa]), __builtin_bit_cast(dwb_bf16x8, bq[nt][pb]),
                                                                        acc[mt][nt], 0, 0, 0);
              }
            }
          if (mt == 0) { if (!ZODD) z_convert(ZBUF); }
          else if (mt == 1) { z_store(ZPOS % NDS, ZODD ? 1 : 0); if (ZODD) z_load(ZBUF, py0 + (t + PB + P + 1) / 2 + 3); }
          else { x_store((SQ + 1) & 1, XS ^ 1); x_load((SQ + 1) & 1, q_lo + t + 3); }
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        const int hs = blk >> 2, mt = blk & 3;
        // the next block's row tile (the next row's first one behind the last block: its slot was written in block 4 .. 6)
        k16_u32x4 an[3];
        if (blk < 7) a_load(an, XS, (blk + 1) & 3); else a_load(an, XS ^ 1, 0);
#pragma unroll
        for (int sum = ORDER; sum >= 0; --sum)                // small products first
#pragma unroll
          for (int pa = 2; pa >= 0; --pa) {
            const int pb = sum - pa;
            if (pb >= 0 && pb <= 2) {
#pragma unroll
              for (int n2 = 0; n2 < 2; ++n2)
                acc[mt][2 * hs + n2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dwb_bf16x8, av[pa]), __builtin_bit_cast(dwb_bf16x8, bq[2 * hs + n2][pb]),
                                                                               acc[mt][2 * hs + n2], 0, 0, 0);
            }
          }
        // this block's share of the staging and of the operand requests
        if (blk == 0) { b_load(SQ, 2); if (!ZODD) z_convert_half(ZBUF, 0); }
        else if (blk == 1) { b_load(SQ, 3); if (!ZODD) z_convert_half(ZBUF, 1); }
        else if (blk == 2) z_store(ZPOS % NDS, ZODD ? 1 : 0);
        else if (blk == 3) { if (ZODD) z_load(ZBUF, py0 + (t + PB + P + 1) / 2 + 3); x_store1((SQ + 1) & 1, XS ^ 1, 0); }
        else if (blk == 4) x_store1((SQ + 1) & 1, XS ^ 1, 1);
        else if (blk == 5) x_store1((SQ + 1) & 1, XS ^ 1, 2);
        else if (blk == 6) { x_load((SQ + 1) & 1, q_lo + t + 3); b_load((SQ + 1) % G::UNR, 0); }
        else b_load((SQ + 1) % G::UNR, 1);
#pragma unroll
        for (int i = 0; i < 2 * NPROD; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
          if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // the next block's 6 transpose reads
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);                          // VALU
          if (i >= 6 && i < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // a column tile's 3 reads
          if (i >= 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);              // LDS writes
          if (i == 1) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);              // loads
          if (i == 2) __builtin_amdgcn_sched_group_barrier(0x020, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pa = 0; pa < 3; ++pa) av[pa] = an[pa];
      }
      }
    };
    for (int t0 = 0; t0 < rows; t0 += G::UNR) {
      if (t0 + 0 < rows) step(std::integral_constant<int, 0>{}, t0 + 0);
      if (t0 + 1 < rows) step(std::integral_constant<int, 1>{}, t0 + 1);
      if (t0 + 2 < rows) step(std::integral_constant<int, 2>{}, t0 + 2);
      if (t0 + 3 < rows) step(std::integral_constant<int, 3>{}, t0 + 3);
      if (t0 + 4 < rows) step(std::integral_constant<int, 4>{}, t0 + 4);
      if (t0 + 5 < rows) step(std::integral_constant<int, 5>{}, t0 + 5);
      if constexpr (G::UNR == 12) {
        if (t0 + 6 < rows) step(std::integral_constant<int, 6>{}, t0 + 6);
        if (t0 + 7 < rows) step(std::integral_constant<int, 7>{}, t0 + 7);
        if (t0 + 8 < rows) step(std::integral_constant<int, 8>{}, t0 + 8);
        if (t0 + 9 < rows) step(std::integral_constant<int, 9>{}, t0 + 9);
        if (t0 + 10 < rows) step(std::integral_constant<int, 10>{}, t0 + 10);
        if (t0 + 11 < rows) step(std::integral_constant<int, 11>{}, t0 + 11);
      }
    }
  }

  // ---- one partial per workgroup: the four waves' accumulators through their own LDS slots, added in wave order
  float* mine = reinterpret_cast<float*>(wvb);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4*>(mine + ((mt * NT + nt) * 64 + lane) * 4) = acc[mt][nt];
  __syncthreads();
  float* part = a.partial + (long)bx * a.pstride;
#pragma unroll
  for (int tile = 0; tile < MT * NT; ++tile) {
    if ((tile & 3) != swave) continue;                        // (wave-uniform: tile i is wave i mod 4's)
    const int mt = tile / NT, nt = tile - mt * NT;
    const int n = 16 * nt + li;
    const bool nvalid = n < KS * NO;
    const int nky = nvalid ? n / NO : 0, no = nvalid ? n % NO : 0;
    f32x4 s = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dwrs_lds) + (tile * 64 + lane) * 4);
#pragma unroll
    for (int v = 1; v < 4; ++v) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dwrs_lds + v * G::WVB) + (tile * 64 + lane) * 4);
      s[0] += o[0]; s[1] += o[1]; s[2] += o[2]; s[3] += o[3];
    }
    if (nvalid && no < a.nout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + 4 * lj + r;
        const int kx = m / CP, c = m - kx * CP;
        if (kx < KS && c < CIN) part[(nky * (KS * CIN) + kx * CIN + c) * a.nout + no] = s[r];
        else if (kx == P && c == CIN && nky == P) part[G::NW / NO * a.nout + no] = s[r];      // the ones channel x the centre tap: sum of dZ = db
      }
    }
  }
}

template <int KSZ, int ORDER>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dw_rs_kernel(const ConvArgsN batch, int units_per_img, int band) {
  conv_dw_rs_body<KSZ, ORDER>(batch, units_per_img, band, blockIdx.x, blockIdx.y);
}

// conv2's (5x5, rows of 32 / 64 pixels) and conv3's (3x3, rows of 16 / 32 / 64) dW, 10 -> 10 channels, pooled dZ (no batch norm)
int conv_dw_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, int* grid, bool* handled);

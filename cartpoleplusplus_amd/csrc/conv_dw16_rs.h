// conv1's weight / bias gradient on the f16 matrix pipes, ONE WAVE PER UNIT ("dw16rs"): v_mfma_f32_16x16x32_f16.
//
// conv_dw16.h's arithmetic contract -- dW = 2^-S (s_c G + t_c T), G the correlation of the RAW f16 pixels with the two f16 pieces of
// dY 2^S, T the same with the ones channel, f32 accumulation -- and its operand layouts (the raw row at a pixel pitch of CP halves read
// through ds_read_b64_tr_b16, dY pieces as [piece][o][lane group][8 halves]), in conv_dw_rs.h's division of labour:
//
//   * a workgroup = one (image, band of rows); its four WAVES = {actor, critic} x {left, right 32-pixel column}.  A wave owns its
//     unit's MT x 4 accumulator tiles of D[m = (kx, c')][n = (ky, o)] (18 channels: 7 x 4 = 112 VGPRs, 56 MFMAs per input row), every A fragment read once
//     per row (conv_dw16.h: once per wave, four times per row and workgroup), NO barrier in the row loop -- the wave stages its own
//     window of the row (its 32 pixels and P on each side, real pixels of the other column where there are any) and its own dY rows;
//   * 2^S is the wave's own (the largest |pooled gradient| of its rows and columns); the column pair of a network is added through
//     LDS after scaling back, left first, then whitened (T through a wave-private table) and written: one partial per workgroup
//     and network, as conv_dw16.h's pair kernel leaves;
//   * the bias gradient is row (kx = P, c' = CIN: the ones channel) x column (ky = P, o).
// 64-wide rows, two f16 pieces (CPP_PRECISION_FAST), networks in pairs that read the same images (actor, critic).  Instances: 18 channels
// (7 x 4 tiles); 9 (4 x 4 tiles = 32 MFMAs per row: measured no faster than conv_dw16.h there, an opt-in of the ablation build).
#pragma once
#include <type_traits>
#include "conv_dw16.h"

template <int CIN_>
struct Dw16RsGeom {
  static constexpr int KS = 5, P = 2, PB = 2, CIN = CIN_, NO = KYO_NO, WC = 32, NPC = 2;
  // channel pitch of a pixel in LDS (halves), as conv_dw16.h's: the channels, the ones channel, rounded to 8 bytes (18 channels: 20,
  // one spare; 9: 12; 6: 8; 3: 4; 12: 16 + 4 -- a pitch of 0 (mod 8) dwords puts the 8 pixel quads of a transpose read on the same banks)
  static constexpr int CP0 = (CIN + 1 + 3) & ~3;
  static constexpr int CP = ((CP0 / 2) % 8 == 0) ? CP0 + 4 : CP0;
  static constexpr bool ODD = (CIN & 1) != 0;                 // pixels are only 2-byte aligned in memory: a raw dword is stored half by half
  static constexpr int MT = (KS * CP + 15) / 16, NT = (KS * NO + 15) / 16;      // 7 x 4 tiles at 18 channels, 4 x 4 at 9
  static constexpr int WPX = WC + 2 * P;                      // pixels of a staged row
  static constexpr int ROWB = 2 * (((CP * WPX + 16 * MT - KS * CP) + 7) & ~7);      // + the m over-read, bytes
  static constexpr int NXS = 2;
  static constexpr int DOST = 80, DPC = NO * DOST;             // dY row: [piece][o][4 lane groups x 16 bytes + skew]
  // (slots padded to 224 mod 256 bytes, which spreads a B read's 16 columns -- two or three ky, i.e. ring slots -- over 16 different bank
  // groups: no change, 3159 / 3160 / 3178 vs 3159 / 3165 / 3150 steps/s alternating on one box; the flat layout stays)
  static constexpr int DSLOT = NPC * DPC;
  static constexpr int NDS = KS + 1;
  static constexpr int WVB = NXS * ROWB + NDS * DSLOT;        // per wave
  static constexpr int SUMB = MT * NT * 64 * 16;              // a network's accumulators on their way to the partial: MT x NT tiles x 64 lanes x 16 bytes
  static constexpr int BASEB = 4 * WVB > 2 * SUMB ? 4 * WVB : 2 * SUMB;      // the rings and the epilogue's sums share the front of the allocation
  static constexpr int TXB = NT * KS * 16 * 4;                // a wave's T table
  static constexpr int LDS_BYTES = (BASEB + 4 * TXB + 2 * CIN * 4 + 15) & ~15;
  static constexpr int NW = KS * KS * CIN * NO;
  static_assert(ROWB % 16 == 0 && WVB % 16 == 0 && BASEB % 16 == 0, "16-byte aligned slots");
  static_assert(MT * NT * 4 <= 112, "the accumulators of a unit: at most 112 registers (18 channels); 30 channels would need 160");
};

// bx = image * nbands + band; by = the pair's first network (networks by and by + 1 read the same images with the same whitening)
template <int CIN_>
__device__ __forceinline__ void conv_dw16_rs_body(const ConvArgsN& batch, const int nbands, const int band, const int bx, const int by) {
  typedef Dw16RsGeom<CIN_> G;
  constexpr int KS = G::KS, P = G::P, PB = G::PB, CIN = G::CIN, NO = G::NO, WC = G::WC, CP = G::CP, MT = G::MT, NT = G::NT, NPC = G::NPC;
  constexpr int ROWB = G::ROWB, DOST = G::DOST, DPC = G::DPC, DSLOT = G::DSLOT, NDS = G::NDS;
  constexpr unsigned BIG = 0x08000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char dw16rs_lds[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lj = lane >> 4;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int net = swave & 1, col = swave >> 1;
  const ConvArgs& a = batch.a[by + net];
  unsigned char* wvb = dw16rs_lds + swave * G::WVB;
  unsigned char* xring = wvb;                                 // [NXS][ROWB]
  unsigned char* dzring = wvb + G::NXS * ROWB;                // [NDS][2 pieces][NO][DOST]
  float* wsc = reinterpret_cast<float*>(dw16rs_lds + G::BASEB + 4 * G::TXB);      // [CIN] scale, [CIN] shift
  const int H = a.H, Hp = H >> 1, W = a.W, Wp = W >> 1, nout = a.nout;
  const int ub = bx / nbands, bd = bx - ub * nbands;
  const int x0 = col * WC;
  const int q_lo = bd * band;
  const int rows = min(band, H - q_lo);                       // (band and q_lo are even)
  const int y0 = q_lo - PB;                                   // dY row of ring position 0 (even: a pooled row's first)
  if (tid < CIN) { wsc[tid] = a.scale[tid]; wsc[CIN + tid] = a.shift[tid]; }

  // ---- the wave's slots: zero; the ones channel (f16 1.0 in channel CIN) of the window's in-image pixels, in both input slots
  for (int i = lane; i < G::WVB / 16; i += 64) reinterpret_cast<k16_u32x4*>(wvb)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
  for (int i = lane; i < G::NXS * G::WPX; i += 64) {
    const int s = i / G::WPX, wx = i - s * G::WPX, x = x0 - P + wx;
    if (x >= 0 && x < W) *reinterpret_cast<unsigned short*>(xring + s * ROWB + 2 * (CP * wx + CIN)) = (unsigned short)0x3C00u;
  }

  // ---- input rows: the raw dwords of the window's pixels.  Even channel count: idx = lane + 64 i = (CIN / 2) wx + cp, a dword = two
  // channels of pixel wx.  Odd: the window's WPX CIN halves are contiguous in the image row and start on a dword (two padding pixels
  // are an even number of halves): dword idx holds halves 2 idx, 2 idx + 1 -- of one pixel or of two neighbours -- and each goes to its
  // own place in the pitched row (a dword is either wholly inside the image or wholly outside: the padding is 2 pixels = 2 CIN halves)
  constexpr bool ODDC = G::ODD;
  constexpr int NXV = ODDC ? (G::WPX * CIN / 2 + 63) / 64 : (G::WPX * (CIN / 2) + 63) / 64;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__half*>((const __half*)a.in + (long)(a.img_slot ? a.img_slot[ub] : ub) * a.in_bstride), 0, H * W * CIN * 2, 0x00020000);
  unsigned xvo[NXV]; uint32_t xdst[NXV]; uint32_t xdst_hi[ODDC ? NXV : 1];
#pragma unroll
  for (int i = 0; i < NXV; ++i) {
    const int idx = lane + 64 * i;
    if (!ODDC) {
      const int wx = idx / (CIN / 2), cp = idx - (CIN / 2) * wx;
      const int x = x0 - P + wx;
      const bool on = idx < G::WPX * (CIN / 2);
      xvo[i] = (on && x >= 0 && x < W) ? (unsigned)((x * CIN + 2 * cp) * 2) : BIG;
      xdst[i] = keep_in_vgpr(lds_addr(xring + (on ? 2 * (CP * wx + 2 * cp) : ROWB - 8)));      // (idle lanes: zeros into the row's tail)
    } else {
      const int h0 = 2 * idx, h1 = 2 * idx + 1;
      const int wx0 = h0 / CIN, wx1 = h1 / CIN;
      const int x = x0 - P + wx0;                                // (both halves' pixels are inside the image or both outside)
      const bool on = idx < G::WPX * CIN / 2;
      xvo[i] = (on && x >= 0 && x < W) ? (unsigned)(((x0 - P) * CIN + h0) * 2) : BIG;
      xdst[i] = keep_in_vgpr(lds_addr(xring + (on ? 2 * (CP * wx0 + (h0 - CIN * wx0)) : ROWB - 8)));
      xdst_hi[i] = keep_in_vgpr(lds_addr(xring + (on ? 2 * (CP * wx1 + (h1 - CIN * wx1)) : ROWB - 6)));
    }
  }
  unsigned xraw[2][NXV];                                      // two rows in flight
  auto x_load = [&](const int buf, const int q) __attribute__((always_inline)) {      // (the row offset in the VGPR: the range check does not see soffset)
    const unsigned ro = q < H ? (unsigned)(q * (W * CIN * 2)) : BIG;
#pragma unroll
    for (int i = 0; i < NXV; ++i) xraw[buf][i] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (int)(xvo[i] + ro), 0, 0);
  };
  auto x_store = [&](const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NXV; ++i) {
      if (!ODDC) lds_store(xdst[i], slot * ROWB, xraw[buf][i]);
      else { lds_store_u16(xdst[i], slot * ROWB, xraw[buf][i] & 0xFFFFu); lds_store_u16(xdst_hi[i], slot * ROWB, xraw[buf][i] >> 16); }
    }
  };

  // ---- dY rows: lane l < 40 owns channel o = l % 10 of lane group g = l / 10: the 16 bytes one lane of the B operand reads -- pixels
  // 16 (g & 1) + 2 (g >> 1) + 4 j + r of the column, i.e. both pixels of the pooled cells x0 / 2 + 8 (g & 1) + (g >> 1) + 2 j, j = 0 .. 3
  const int zg = lane / NO, zo = lane - zg * NO;
  const bool zon = lane < 4 * NO;
  const __amdgpu_buffer_rsrc_t dp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy.dpool + (long)ub * a.dy.dpool_bstride), 0,
                                                                           Hp * Wp * NO * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t am_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dy.amax + (long)ub * Hp * Wp * NO), 0,
                                                                           Hp * Wp * NO, 0x00020000);
  const unsigned zoff = zon ? (unsigned)((x0 / 2 + 8 * (zg & 1) + (zg >> 1)) * NO + zo) : BIG;      // element offset of cell j = 0 in a pooled row; cell j: + 2 j NO
  const uint32_t zdst = keep_in_vgpr(lds_addr(dzring + (zon ? zo * DOST + zg * 16 : DOST - 16)));      // (idle lanes: zeros into the skew)

  // ---- 2^S: the largest |pooled gradient| of the pooled rows and cells this wave stages lands in [2^14, 2^15)
  float sc, inv;
  {
    const int pylo = max(0, (q_lo - P) >> 1), pyhi = min(Hp - 1, (q_lo + rows - 1 + P) >> 1);
    float vmax = 0.f;
    if (a.dy.imax) {                                          // the image's bound, left by the kernel that wrote the gradient (round 6: the scan below was 3.3 us of the launch)
      const f32x4 im = *reinterpret_cast<const f32x4*>(a.dy.imax + (long)ub * DX_IMAX_SLOTS), im2 = *reinterpret_cast<const f32x4*>(a.dy.imax + (long)ub * DX_IMAX_SLOTS + 4);
      vmax = fmaxf(fmaxf(fmaxf(im[0], im[1]), fmaxf(im[2], im[3])), fmaxf(fmaxf(im2[0], im2[1]), fmaxf(im2[2], im2[3])));
    }
    // (nine pooled rows = 36 loads in flight per trip: a 32-row band's 18 pooled rows are two round trips -- four rows per trip were five,
    // in front of everything else the wave does; rows behind pyhi: out of range, zeros)
    constexpr int PSR = 9;
    for (int py = a.dy.imax ? pyhi + 1 : pylo; py <= pyhi; py += PSR) {
      float t[PSR][4];
#pragma unroll
      for (int u = 0; u < PSR; ++u) {
        const unsigned ro = py + u <= pyhi ? (unsigned)((py + u) * (Wp * NO)) : BIG;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(dp_rsrc, (int)((zoff + ro + (unsigned)(2 * j * NO)) * 4u), 0, 0));
      }
#pragma unroll
      for (int u = 0; u < PSR; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) vmax = fmaxf(vmax, fabsf(t[u][j]));
    }
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    int S = 0;
    if (vmax > 0.f && vmax < 3.0e38f) S = 14 - ilogbf(vmax);
    S = S > 100 ? 100 : (S < -100 ? -100 : S);
    sc = ldexpf(1.f, S); inv = ldexpf(1.f, -S);
  }

  float zrg[2][4]; unsigned zrc[2][4];                        // two pooled rows in flight (a row is requested three steps before its conversion)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) { zrg[b][j] = 0.f; zrc[b][j] = 0u; }
  unsigned zp01[NPC] = {0u, 0u}, zp23[NPC] = {0u, 0u};        // the pieces of cells (0, 1) and (2, 3), packed
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
    unsigned short pc[2][NPC];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float g = (zrc[buf][2 * hf + j] & POOL_ACTIVE) ? zrg[buf][2 * hf + j] : 0.f;
      const float v = g * sc;
      const _Float16 h = (_Float16)v;
      const _Float16 m = (_Float16)(v - (float)h);
      pc[j][0] = __builtin_bit_cast(unsigned short, h); pc[j][1] = __builtin_bit_cast(unsigned short, m);
    }
#pragma unroll
    for (int p = 0; p < NPC; ++p) {
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
  auto z_store = [&](const int slot, const int ry) __attribute__((always_inline)) {      // dY row 2 py + ry of the converted pooled row
#pragma unroll
    for (int p = 0; p < NPC; ++p)
      lds_store(zdst, slot * DSLOT + p * DPC, (k16_u32x4){zp01[p] & zm01[2 * ry], zp23[p] & zm23[2 * ry], zp01[p] & zm01[2 * ry + 1], zp23[p] & zm23[2 * ry + 1]});
  };

  // ---- MFMA operands (pixel dealing and transpose reads as conv_dw16.h)
  const int tj = (lane >> 2) & 3, tq = lane & 3;
  const uint32_t aadr = keep_in_vgpr(lds_addr(xring + 2 * (CP * (16 * (lj & 1) + 4 * tj + 2 * (lj >> 1)) + 4 * tq)));
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

  {
    // ---- prologue: dY ring positions 0 .. PB + P, input row q_lo; two more of each in flight
    const int py0 = y0 >> 1;                                  // (-1 for the first band: zeros)
    z_load(0, py0); z_load(1, py0 + 1);
    x_load(0, q_lo); x_load(1, q_lo + 1);
#pragma unroll
    for (int d = 0; d <= PB + P; ++d) {                       // (the position behind the last one is stored by the first step)
      if ((d & 1) == 0) { z_convert((d / 2) % 2); z_load((d / 2) % 2, py0 + d / 2 + 2); }
      z_store(d % NDS, d & 1);
    }
    x_store(0, 0);
    x_load(0, q_lo + 2);
    __builtin_amdgcn_sched_barrier(0);

    // Step t (sq = t mod 12, compile time): multiply input row q_lo + t (slot t & 1) with dY positions t .. t + 2 P; meanwhile dY position
    // t + 2 P + 1 (pooled row py0 + (t + 5) / 2, buffer ((t + 5) / 2) mod 2) and input row q_lo + t + 1 go to LDS, and the loads of
    // pooled row py0 + (t + 5) / 2 + 2 (every second step) and of input row q_lo + t + 3 leave -- a seventh of it behind each row tile
    k16_u32x4 bq[NT][NPC];
    auto a_load = [&](k16_u32x4& dst, const int xs, const int mt) __attribute__((always_inline)) {
      const int off = xs * ROWB + mt * 32;
      const dw16_v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)off)));
      const dw16_v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)(off + 2 * CP))));
      const dw16_u32x2 u0 = __builtin_bit_cast(dw16_u32x2, r0), u1 = __builtin_bit_cast(dw16_u32x2, r1);
      dst = (k16_u32x4){u0.x, u0.y, u1.x, u1.y};
    };
    auto step = [&](auto sqtag, const int t) __attribute__((always_inline)) {
      constexpr int SQ = decltype(sqtag)::value;
      constexpr int XS = SQ & 1;
      constexpr int ZPOS = SQ + PB + P + 1;
      constexpr bool ZODD = (ZPOS & 1) != 0;
      constexpr int ZBUF = (ZPOS / 2) % 2;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int sl = (SQ + PB + P) % NDS - bky[nt]; sl = sl < 0 ? sl + NDS : sl;      // slot of this lane's column: ky differs per lane
        const uint32_t ad = bbase[nt] + (uint32_t)(sl * DSLOT);
#pragma unroll
        for (int p = 0; p < NPC; ++p) bq[nt][p] = lds_load<k16_u32x4>(ad, p * DPC);
      }
      k16_u32x4 av;
      a_load(av, XS, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        k16_u32x4 an = av;
        if (mt + 1 < MT) a_load(an, XS, mt + 1);
#pragma unroll
        for (int pc = NPC - 1; pc >= 0; --pc)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bq[nt][pc]), acc[mt][nt], 0, 0, 0);
        // (the six staging tasks of a step, one behind each of the first row tiles' MFMAs; fewer than six row tiles: the rest behind the last)
        constexpr int TL = MT - 1;
        if (mt == (0 < TL ? 0 : TL)) { if (!ZODD) z_convert_half(ZBUF, 0); }
        if (mt == (1 < TL ? 1 : TL)) { if (!ZODD) z_convert_half(ZBUF, 1); }
        if (mt == (2 < TL ? 2 : TL)) z_store(ZPOS % NDS, ZODD ? 1 : 0);
        if (mt == (3 < TL ? 3 : TL)) { if (ZODD) z_load(ZBUF, py0 + (t + PB + P + 1) / 2 + 2); }
        if (mt == (4 < TL ? 4 : TL)) x_store((SQ + 1) & 1, XS ^ 1);
        if (mt == (5 < TL ? 5 : TL)) x_load((SQ + 1) & 1, q_lo + t + 3);
        av = an;
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    // (12 steps per unrolled block: the slots rotate with period 6, the two pooled-row buffers with period 4)
    for (int t0 = 0; t0 < rows; t0 += 12) {
      if (t0 + 0 < rows) step(std::integral_constant<int, 0>{}, t0 + 0);
      if (t0 + 1 < rows) step(std::integral_constant<int, 1>{}, t0 + 1);
      if (t0 + 2 < rows) step(std::integral_constant<int, 2>{}, t0 + 2);
      if (t0 + 3 < rows) step(std::integral_constant<int, 3>{}, t0 + 3);
      if (t0 + 4 < rows) step(std::integral_constant<int, 4>{}, t0 + 4);
      if (t0 + 5 < rows) step(std::integral_constant<int, 5>{}, t0 + 5);
      if (t0 + 6 < rows) step(std::integral_constant<int, 6>{}, t0 + 6);
      if (t0 + 7 < rows) step(std::integral_constant<int, 7>{}, t0 + 7);
      if (t0 + 8 < rows) step(std::integral_constant<int, 8>{}, t0 + 8);
      if (t0 + 9 < rows) step(std::integral_constant<int, 9>{}, t0 + 9);
      if (t0 + 10 < rows) step(std::integral_constant<int, 10>{}, t0 + 10);
      if (t0 + 11 < rows) step(std::integral_constant<int, 11>{}, t0 + 11);
    }
  }

  // ---- one partial per workgroup and network: the left column's accumulators (x 2^-S) through LDS, the right column's wave adds its
  // own and holds the sums: T (rows m = CP kx + CIN) through a wave-private table, then dW = s_c G + t_c T
  __syncthreads();                                            // (everybody is done with the rings: the buffers below lie over them)
  float* sum = reinterpret_cast<float*>(dw16rs_lds + net * G::SUMB);
  if (col == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(sum + ((mt * NT + nt) * 64 + lane) * 4) = (f32x4){acc[mt][nt][0] * inv, acc[mt][nt][1] * inv, acc[mt][nt][2] * inv, acc[mt][nt][3] * inv};
  }
  __syncthreads();
  if (col == 0) return;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(sum + ((mt * NT + nt) * 64 + lane) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][nt][r] = o[r] + acc[mt][nt][r] * inv;
    }
  float* tx = reinterpret_cast<float*>(dw16rs_lds + G::BASEB + swave * G::TXB);      // [NT][KS][16]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int m = CP * kx + CIN;                            // compile-time
      if (lj == ((m & 15) >> 2)) tx[(nt * KS + kx) * 16 + li] = acc[m >> 4][nt][m & 3];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float* part = a.partial + (long)bx * a.pstride;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = 16 * nt + li;
    const bool nvalid = n < KS * NO;
    const int nky = nvalid ? n / NO : 0, no = nvalid ? n % NO : 0;
    if (nvalid && no < nout) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * mt + 4 * lj + r;
          const int kx = m / CP, c = m - kx * CP;
          if (kx < KS && c < CIN) {
            const float t = tx[(nt * KS + kx) * 16 + li];
            part[(nky * (KS * CIN) + kx * CIN + c) * nout + no] = wsc[c] * acc[mt][nt][r] + wsc[CIN + c] * t;
          } else if (kx == P && c == CIN && nky == P) part[G::NW / NO * nout + no] = acc[mt][nt][r];      // the ones channel x the centre tap: sum of dY = db
        }
    }
  }
}

// with the next minibatch's sample + statistics pass behind it in the same grid (conv1_dw_gather.hip)
int conv_dw16_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool dense, const ConvArgsN& a, int* grid, bool* handled);

// conv1 weight / bias gradient on the f16 matrix pipes with f32-grade operands ("dw16"): v_mfma_f32_16x16x32_f16.
//
//   dW[ky][kx][c][o] = sum_{b,q,x} xw[b, q, x + kx - P, c] * dY[b, q - ky + P, x, o],   xw = x * s_c + t_c inside the image, 0 in the padding
//                    = s_c * G[ky][kx][c][o] + t_c * T[ky][kx][o]
//   G = the same correlation with the RAW f16 pixel x (exact in f16: replay_memory.py:32), T = the same with the "ones" channel
//   (1 inside the image) -- both come out of ONE MFMA stream whose A rows are (kx, c'), c' = 0..CIN-1 the pixel channels, CIN the
//   ones channel.  dY (f32) is split into F16_PIECES f16 pieces dY 2^S = h + m (+ l) (S from the largest |pooled gradient| the
//   workgroup will see; conv_k16.h: two pieces = dY to within one f32 ulp, release; three = dY itself, exact build), every
//   f16 x f16 product is exact, the accumulators are f32: the f32-MFMA kernel's arithmetic (conv_dw_kyo.h), two / three 16-cycle
//   MFMAs per 32 pixels instead of eight 32-cycle ones.
//
// Formulation as conv_dw_kyo.h: per input row q,  D[m = (kx,c'), n = (ky,o)] += A[m, pixel] * B[pixel, n]; wave w owns column
// tile w of (ky,o) and all MT row tiles (30 channels: half the row tiles of two column tiles, Dw16Geom::SPLIT); units = (image, band of rows); one partial per workgroup (whitening applied to it:
// s_c G + t_c T) for conv_dw_reduce_kernel.
//
// The contraction runs over PIXELS, but an image row is channel-contiguous.  The row is copied raw into LDS with a pixel
// pitch of CP = CIN + 2 halves (8-byte aligned pixels; channel CIN holds the constant 1, CIN + 1 is zero), so that
// A[m = CP kx + c][x] = row[CP x + m] is flat in m, and ds_read_b64_tr_b16 delivers the transposed fragment: lane quad j of a
// 16-lane group supplies the address of rows 4q..4q+3 of one pixel, lane i receives row i of four pixels.  The pixels of a
// 32-pixel chunk are dealt to (lane group g, read r, quad j) as x = 16 (g & 1) + 4 j + 2 (g >> 1) + r: the 8 quads of a
// 32-lane half then hit 8 disjoint bank octets (pitch 10 dwords), and the two reads of a lane are the two pixels of one
// pooled cell.  dY rows sit in LDS as [piece][o][chunk][g][8 halves] in that same pixel order (16-byte B operands).
#pragma once
#include "conv_k16.h"

typedef short dw16_v4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef unsigned dw16_u32x2 __attribute__((ext_vector_type(2)));

template <int CIN, int KS, int NCHK, int NPCS = F16_PIECES>
struct Dw16Geom {
  static constexpr int P = KS / 2, NO = KYO_NO, NPC = NPCS;
  // channel pitch of a pixel in LDS (halves): CIN + the ones channel, rounded to 8 bytes; a pitch of 0 (mod 8) dwords would
  // put the 8 pixel quads of a transpose read on the same banks, so those get 4 more halves (30 channels -> 36)
  static constexpr int CP0 = (CIN + 1 + 3) & ~3;
  static constexpr int CP = ((CP0 / 2) % 8 == 0) ? CP0 + 4 : CP0;
  static constexpr int MROWS = KS * CP, MT = (MROWS + 15) / 16;
  // How a workgroup's four waves divide the MT x 4 accumulator tiles: MT x 1 each (wave w owns column tile w), or -- SPLIT, 30 channels:
  // 12 row tiles -- (MT / 2) x 2 each (waves 0, 1: column tiles 0, 1; waves 2, 3: 2, 3; even waves the lower half of the row tiles).  Per
  // 32-pixel chunk a wave then reads 6 A tiles + 2 x NPC B tiles from LDS instead of 12 + NPC: 10 KB instead of 14 KB for the same 24
  // MFMAs, and at two workgroups per CU the LDS reads of that instance take as long as its MFMAs (probe: profiles/experiments/r06_dw16_split_probe.*)
  static constexpr bool SPLIT = MT >= 10 && MT % 2 == 0;
  static constexpr int NCT = SPLIT ? 2 : 1, MTW = SPLIT ? MT / 2 : MT;
  static constexpr int KROW = KS * CIN;
  static constexpr int WPAD = 32 * NCHK;
  static constexpr int ROWH = ((CP * (WPAD + KS - 1) + 16 * MT - MROWS) + 7) & ~7;     // halves per staged input row (+ m over-read)
  static constexpr int ROWB = ROWH * 2;
  static constexpr int DOST = 64 * NCHK + 32;                 // bytes per (piece, o): NCHK chunks of 64 + skew
  static constexpr int DPC = NO * DOST;
  static constexpr int DSLOT = ((NPC * DPC - 192 + 255) / 256) * 256 + 192;            // = 192 (mod 256): B reads 1.17 accesses per bank quad
  static constexpr int RING_IN = 3, RING_DY = 6, UNROLL = 6;
  // dY staging: a thread's TASK is two pooled cells (px, o), (px + 2, o) whose f16 pieces are neighbours in the LDS layout (one 4-byte
  // store per piece and x parity instead of two 2-byte ones): NO x NCHK x 4 lane groups x 2 pairs tasks per pooled row
  static constexpr int NTASK = (NO * NCHK * 8 + CONV_THREADS - 1) / CONV_THREADS;
  static constexpr int NCELL = 2 * NTASK;                     // pooled cells of a row per thread
  static constexpr int NVIN = (CIN % 2 ? WPAD * CIN : WPAD * CIN / 2) / CONV_THREADS + 1;   // dwords (odd CIN: halves) of an input row per thread
  static constexpr int IN_BYTES = RING_IN * ROWB, DY_BYTES = RING_DY * DSLOT;
  static constexpr int LDS_BYTES = ((IN_BYTES + 15) & ~15) + DY_BYTES + 64;       // (the epilogue scratch reuses the dY ring)
  static constexpr int LDS_BYTES2 = ((IN_BYTES + 15) & ~15) + 2 * DY_BYTES + 64;  // two networks per workgroup: one input ring, two dY rings
  static_assert(DY_BYTES >= CONV_THREADS * NCELL * 4 + 4 * KS * 16 * 4 + 2 * CIN * 4, "epilogue scratch fits the dY ring");
};

#ifndef DW16_WGS
#define DW16_WGS 3
#endif
#ifndef DW16_CAP
#define DW16_CAP 4
#endif

// DENSE: dY comes as dense f32 rows (a.dy_dense: batch norm's dz) instead of being rebuilt from the pooled gradient
// (bx, by, gx): the workgroup's place in a (gx, networks) grid (its own launch, or a slice of a shared one: conv1_dw_gather.hip)
// NNET = 2: ONE workgroup serves networks by and by + 1, which read the SAME images (the actor and the critic both take state_1:
// ddpg_cartpole.py:333-334) -- the raw row is staged once, every A fragment is read from LDS once and multiplied with both networks' dY
// (two dY rings, two accumulator sets, two partials).  Half the input staging and half the A reads per MFMA; the joint launch is 512
// workgroups = one resident round instead of 1024 on 768 slots.
template <int CIN, int KS, int NCHK, bool DENSE = false, int NNET = 1, int NPCS = F16_PIECES>
__device__ __forceinline__ void conv_dw16_body(const ConvArgsN& batch, int units_per_img, int band, const int bx, const int by, const int gx) {
  typedef Dw16Geom<CIN, KS, NCHK, NPCS> G;
  constexpr int P = G::P, NO = G::NO, CP = G::CP, NPC = G::NPC, ROWB = G::ROWB, DSLOT = G::DSLOT;
  static_assert((KS * NO + 15) / 16 == 4, "one column tile per wave");
  static_assert(NNET == 1 || NNET == 2, "one or two networks per workgroup");
  static_assert(NNET == 1 || !DENSE, "the two-network workgroup rebuilds dY from the pooled gradient");
  constexpr bool ODD = (CIN & 1) != 0;               // pixels are only 2-byte aligned in memory: rows are staged half by half
  constexpr int DYB = G::DY_BYTES;                   // network k's dY ring sits k * DYB bytes behind network 0's
#ifdef DW16_CLOCK
  const unsigned long long ce0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long cpro = 0, cloop = 0;
#endif
  const ConvArgs& a = batch.a[by];                   // geometry, the images and their whitening: shared by the NNET networks
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* inring = lds_raw;                                         // [3][ROWB]
  unsigned char* dyring = lds_raw + ((G::IN_BYTES + 15) & ~15);            // [NNET][6][DSLOT]
  float* red = reinterpret_cast<float*>(dyring + NNET * DYB);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;
  const int units = a.B * units_per_img;

  // (the epilogue's whitening scale / shift are requested here: in the epilogue the load was one more L2 round trip per workgroup)
  float wsc_s = 0.f, wsc_t = 0.f;
  if (tid < CIN) { wsc_s = a.scale[tid]; wsc_t = a.shift[tid]; }
  float sc[NNET], inv[NNET];                        // 2^S, 2^-S per network: set by the pre-scan below, once the first loads are in flight

  // ---- input row staging: raw dwords (two channels; odd CIN: single halves) of the f16 row -> pixel pitch CP
  unsigned sv[G::NVIN];
  bool sact[G::NVIN];
  uint32_t sdst[G::NVIN];
  constexpr int EPX = ODD ? CIN : CIN / 2;            // staging elements per pixel
#pragma unroll
  for (int i = 0; i < G::NVIN; ++i) {
    const int d = tid + CONV_THREADS * i;
    const int x = d / EPX, w = d - x * EPX;
    sact[i] = d < W * EPX;
    sdst[i] = keep_in_vgpr(lds_addr(inring + 2 * (CP * (x + P) + (ODD ? w : 2 * w))));
  }
  const int rowbytes = W * CIN * 2;
  unsigned sv2[G::NVIN];                              // (the unit prologue has two rows in flight)
  auto in_load_to = [&](unsigned (&dst)[G::NVIN], const __amdgpu_buffer_rsrc_t& rs, int q) {
#pragma unroll
    for (int i = 0; i < G::NVIN; ++i)
      if (sact[i]) {
        if (ODD) dst[i] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, (tid + CONV_THREADS * i) * 2, q * rowbytes, 0);
        else dst[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (tid + CONV_THREADS * i) * 4, q * rowbytes, 0);
      }
  };
  auto in_store_from = [&](const unsigned (&src)[G::NVIN], int slot) {
#pragma unroll
    for (int i = 0; i < G::NVIN; ++i)
      if (sact[i]) {
        if (ODD) lds_store(sdst[i], slot * ROWB, (unsigned short)src[i]);
        else lds_store(sdst[i], slot * ROWB, src[i]);
      }
  };
  auto in_load = [&](const __amdgpu_buffer_rsrc_t& rs, int q) { in_load_to(sv, rs, q); };
  auto in_store = [&](int slot) { in_store_from(sv, slot); };

  // ---- dY staging: a thread owns pooled cells idx = px * nout + o of a pooled row (the same cells for every row and network); the
  // masked gradient is scaled, split into three f16 pieces and written to both image rows of the pooled row.
  // pixel x = 32 ch + w sits in lane group g = 2 ((w >> 1) & 1) + (w >> 4), element e = 4 (w & 1) + ((w >> 2) & 3)
  constexpr int NCELL = G::NCELL, NTASK = G::NTASK;
  bool cact[NCELL], tact[NTASK];
  uint32_t cdst[NTASK];
  int cvo[NCELL];                                    // byte offset of the cell's f32 in a pooled row (pool / dpool; / 4: arg-max code)
  float dbsum[NNET][NCELL];
  // per task and network, for the pooled row in work: the three f16 pieces of its two cells packed (low half = cell 0), and for each
  // (image-row parity ry, x parity s) the 32-bit mask that keeps a cell's half where its arg-max code is 2 ry + s
  unsigned tpc[NNET][NTASK][NPC], tmask[NNET][NTASK][4];
#pragma unroll
  for (int tk = 0; tk < NTASK; ++tk) {
    // task T: o fastest (neighbouring lanes load neighbouring floats), then the pair, the lane group, the chunk
    const int T = tid + CONV_THREADS * tk;
    const int o = T % nout, rest = T / nout;
    const int jp = rest & 1, g = (rest >> 1) & 3, ch = rest >> 3;
    tact[tk] = ch < NCHK;
    // lane group g of chunk ch holds pixels w = 16 (g & 1) + 2 (g >> 1) + {0, 1} + 4 j, j = 0..3: pooled cells px0 + 2 j
    const int px0 = 16 * ch + 8 * (g & 1) + (g >> 1);
    cdst[tk] = keep_in_vgpr(lds_addr(dyring + (tact[tk] ? o * G::DOST + ch * 64 + g * 16 + (2 * jp) * 2 : 0)));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 2 * tk + j;
      const int px = px0 + 2 * (2 * jp + j);
      cact[c] = tact[tk] && px < Wp;
      cvo[c] = cact[c] ? (px * nout + o) * 4 : 0;
#pragma unroll
      for (int k = 0; k < NNET; ++k) dbsum[k][c] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc) tpc[k][tk][pc] = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) tmask[k][tk][q] = 0u;
    }
  }
  float rdv[NNET][3][NCELL];   // (set 2: only the unit prologue, so that its three requests are in flight together)
  int rcd[NNET][3][NCELL];
  __amdgpu_buffer_rsrc_t rd[NNET], rc[NNET];
  auto dy_issue = [&](int py, int set) {
    const bool rowok = py >= 0 && py < Hp;           // uniform
#pragma unroll
    for (int k = 0; k < NNET; ++k)
#pragma unroll
      for (int c = 0; c < NCELL; ++c) {
        rdv[k][set][c] = 0.f; rcd[k][set][c] = 0;
        if (rowok && cact[c]) {
          const int vo = cvo[c], so = py * Wp * nout * 4;
          rdv[k][set][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd[k], vo, so, 0));
          rcd[k][set][c] = __builtin_amdgcn_raw_buffer_load_b8(rc[k], vo >> 2, so >> 2, 0);
        }
      }
  };
  auto dy_conv = [&](int set, bool count) {
#pragma unroll
    for (int k = 0; k < NNET; ++k)
#pragma unroll
      for (int tk = 0; tk < NTASK; ++tk) {
        unsigned pk[2][NPC];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = 2 * tk + j;
          const float g = (rcd[k][set][c] & 4) ? rdv[k][set][c] : 0.f;      // bit 2 of the code byte: the pooled output was > 0 (conv_kyo.h: POOL_ACTIVE)
          if (count) dbsum[k][c] += g;
          const float v = g * sc[k];
          const _Float16 h = (_Float16)v;
          const float r1 = v - (float)h;
          const _Float16 m = (_Float16)r1;
          const _Float16 l = (_Float16)(r1 - (float)m);
          pk[j][0] = __builtin_bit_cast(unsigned short, h);
          pk[j][1] = __builtin_bit_cast(unsigned short, m);
          if (NPC > 2) pk[j][NPC - 1] = __builtin_bit_cast(unsigned short, l);
        }
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) tpc[k][tk][pc] = pk[0][pc] | (pk[1][pc] << 16);
        const int c0 = rcd[k][set][2 * tk] & 3, c1 = rcd[k][set][2 * tk + 1] & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) tmask[k][tk][q] = (c0 == q ? 0x0000FFFFu : 0u) | (c1 == q ? 0xFFFF0000u : 0u);
      }
  };
  auto dy_store = [&](int slot, int ry) {            // image row with parity ry of the pooled row -> ring slot (of every network)
#pragma unroll
    for (int k = 0; k < NNET; ++k)
#pragma unroll
      for (int tk = 0; tk < NTASK; ++tk) {
        if (tact[tk]) {
#pragma unroll
          for (int s = 0; s < 2; ++s)                // x parity: elements e and e + 4 of the 16-byte group
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc)
              lds_store(cdst[tk], k * DYB + slot * DSLOT + pc * G::DPC + 8 * s, tpc[k][tk][pc] & tmask[k][tk][2 * ry + s]);
        }
      }
  };

  // ---- dense dY rows (DENSE): a thread owns elements idx = x * nout + o of a row; scaled, split, same LDS layout
  constexpr int NDC = DENSE ? (G::WPAD * NO + CONV_THREADS - 1) / CONV_THREADS : 1;
  bool dact[NDC]; uint32_t ddst[NDC]; float dreg[NDC];
#pragma unroll
  for (int c = 0; c < NDC; ++c) {
    const int idx = tid + CONV_THREADS * c;
    const int x = idx / nout, o = idx - x * nout;
    dact[c] = DENSE && idx < W * nout;
    const int ch = x >> 5, w = x & 31;
    const int g = 2 * ((w >> 1) & 1) + (w >> 4), e = 4 * (w & 1) + ((w >> 2) & 3);
    ddst[c] = keep_in_vgpr(lds_addr(dyring + (o < NO ? o : 0) * G::DOST + ch * 64 + g * 16 + e * 2));
    dreg[c] = 0.f;
  }
  auto dense_load = [&](const __amdgpu_buffer_rsrc_t& rs, int y) {
    const bool rowok = y >= 0 && y < H;               // uniform
#pragma unroll
    for (int c = 0; c < NDC; ++c) {
      dreg[c] = 0.f;
      if (rowok && dact[c]) dreg[c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (tid + CONV_THREADS * c) * 4, y * W * nout * 4, 0));
    }
  };
  auto dense_store = [&](int slot) {
#pragma unroll
    for (int c = 0; c < NDC; ++c) {
      if (dact[c]) {
        const float v = dreg[c] * sc[0];
        const _Float16 h = (_Float16)v;
        const float r1 = v - (float)h;
        const _Float16 m = (_Float16)r1;
        const _Float16 l = (_Float16)(r1 - (float)m);
        lds_store(ddst[c], slot * DSLOT, __builtin_bit_cast(unsigned short, h));
        lds_store(ddst[c], slot * DSLOT + G::DPC, __builtin_bit_cast(unsigned short, m));
        if (NPC > 2) lds_store(ddst[c], slot * DSLOT + 2 * G::DPC, __builtin_bit_cast(unsigned short, l));
      }
    }
  };

  // ---- MFMA operands
  const int tj = (lane >> 2) & 3, tq = lane & 3;
  constexpr int NCT = G::NCT, MTW = G::MTW;          // column tiles / row tiles per wave (Dw16Geom::SPLIT)
  const int wcol = G::SPLIT ? (wave >> 1) : wave, mt0 = G::SPLIT ? (wave & 1) * MTW : 0;      // first column tile = NCT wcol; first row tile
  const uint32_t aadr = keep_in_vgpr(lds_addr(inring + 2 * (CP * (16 * (lj & 1) + 4 * tj + 2 * (lj >> 1)) + 4 * tq) + mt0 * 32));
  bool nvalid[NCT]; int nky[NCT], no[NCT];
  uint32_t badr[NCT][G::UNROLL];
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const int n = 16 * (NCT * wcol + j) + li;
    nvalid[j] = n < KS * NO;
    nky[j] = nvalid[j] ? n / NO : 0; no[j] = nvalid[j] ? n % NO : 0;
#pragma unroll
    for (int sq = 0; sq < G::UNROLL; ++sq) {
      const int slot = (sq - nky[j] + P + G::RING_DY) % G::RING_DY;       // ring slot of dY position t - ky + P, t = sq (mod 6)
      badr[j][sq] = keep_in_vgpr(lds_addr(dyring + slot * DSLOT + no[j] * G::DOST + lj * 16));
    }
  }
  f32x4 acc[NNET][NCT][MTW];
#pragma unroll
  for (int k = 0; k < NNET; ++k)
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) acc[k][j][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- per-unit state and the requests a unit opens with
  int ub = 0, q_lo = 0, rows = 0, y0 = 0;
  __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in), 0, 0, 0x00020000);
  auto unit_begin = [&](int unit) {
    ub = unit / units_per_img;
    q_lo = (unit - ub * units_per_img) * band;
    rows = min(band, H - q_lo);                        // band and q_lo are even
    in_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__half*>((const __half*)a.in + (long)(a.img_slot ? a.img_slot[ub] : ub) * a.in_bstride), 0, H * rowbytes, 0x00020000);
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
      const ConvArgs& ak = batch.a[by + k];
      rd[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ak.dy.dpool + (long)ub * ak.dy.dpool_bstride), 0, Hp * Wp * nout * 4, 0x00020000);
      rc[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(ak.dy.amax + (long)ub * Hp * Wp * nout), 0, Hp * Wp * nout, 0x00020000);
    }
    y0 = q_lo - P;                                     // position d of the band's stream <-> image row q_lo - P + d (conv_dw_kyo.h)
  };
  // (the first two input rows and the first three pooled dY rows are requested together: one L2 round trip, not five)
  auto unit_requests = [&]() {
    if (0 < rows) in_load(in_rs, q_lo);
    if (1 < rows) in_load_to(sv2, in_rs, q_lo + 1);
    dy_issue(y0 >> 1, 0);
    dy_issue((y0 >> 1) + 1, 1);
    dy_issue((y0 >> 1) + 2, 2);
  };
  // The workgroup's first unit sends its requests NOW, in front of the set-up below (ring zeroing, scale pre-scan, two barriers:
  // 5 us by the in-kernel clock): the unit prologue used to wait a full round trip (4 us) for them after the set-up.
  bool primed = false;
  if (!DENSE && bx < units) { unit_begin(bx); unit_requests(); primed = true; }

  // ---- zero the rings; the ones channel of the in-image pixels of every input slot (staging never touches it)
  for (int i = tid; i < (int)(((G::IN_BYTES + 15) & ~15) + NNET * DYB) / 16; i += CONV_THREADS)
    reinterpret_cast<float4*>(lds_raw)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  for (int i = tid; i < G::RING_IN * W; i += CONV_THREADS) {
    const int s = i / W, x = i - s * W;
    *reinterpret_cast<unsigned short*>(inring + s * ROWB + 2 * (CP * (x + P) + CIN)) = (unsigned short)0x3C00u;      // 1.0
  }

  // ---- 2^S: the largest |pooled gradient| among the pooled rows this workgroup's units touch lands in [2^14, 2^15) (per network)
  {
    float vmax[NNET];
#pragma unroll
    for (int k = 0; k < NNET; ++k) vmax[k] = 0.f;
    for (int unit = bx; unit < units; unit += gx) {
      const int b = unit / units_per_img;
      const int q_lo = (unit - b * units_per_img) * band;
      const int rows = min(band, H - q_lo);
      if (DENSE) {
        const int r0 = max(0, q_lo - P), r1 = min(H - 1, q_lo + rows - 1 + P);
        const float* dp = a.dy_dense + (long)b * a.dy_dense_bstride;
        const int e1 = (r1 + 1) * W * nout;
        int e = r0 * W * nout + tid;
        for (; e + 7 * CONV_THREADS < e1; e += 8 * CONV_THREADS) {      // 8 loads in flight (a 1-load loop pays the latency per trip)
          float t[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = dp[e + u * CONV_THREADS];
#pragma unroll
          for (int u = 0; u < 8; ++u) vmax[0] = fmaxf(vmax[0], fabsf(t[u]));
        }
        for (; e < e1; e += CONV_THREADS) vmax[0] = fmaxf(vmax[0], fabsf(dp[e]));
      } else if (a.dy.imax) {
        // the image's bound, left by the kernel that wrote the gradient (conv2's dX on conv_dx_rs.h; round 6): eight floats instead of a
        // scan of the unit's pooled rows in front of everything else the workgroup does
#pragma unroll
        for (int k = 0; k < NNET; ++k) {
          const float* ip = batch.a[by + k].dy.imax + (long)b * DX_IMAX_SLOTS;
          const f32x4 im = *reinterpret_cast<const f32x4*>(ip), im2 = *reinterpret_cast<const f32x4*>(ip + 4);
          vmax[k] = fmaxf(vmax[k], fmaxf(fmaxf(fmaxf(im[0], im[1]), fmaxf(im[2], im[3])), fmaxf(fmaxf(im2[0], im2[1]), fmaxf(im2[2], im2[3]))));
        }
      } else {
        const int py0 = max(0, (q_lo - P) >> 1), py1 = min(Hp - 1, (q_lo + rows - 1 + P) >> 1);
        const int e1 = (py1 + 1) * Wp * nout;
        // the rows come from another kernel's L2 (1.5-2 us a trip): 24 loads in flight per trip and network -- one trip for a half image
        // of the headline shape -- through descriptors that end at e1 (reads past it return 0)
        __amdgpu_buffer_rsrc_t rs[NNET];
#pragma unroll
        for (int k = 0; k < NNET; ++k)
          rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(batch.a[by + k].dy.dpool + (long)b * batch.a[by + k].dy.dpool_bstride), 0, e1 * 4, 0x00020000);
        for (int e = py0 * Wp * nout + tid; e < e1; e += 24 * CONV_THREADS) {
          float t[NNET][24];
#pragma unroll
          for (int k = 0; k < NNET; ++k)
#pragma unroll
            for (int u = 0; u < 24; ++u) t[k][u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs[k], (e + u * CONV_THREADS) * 4, 0, 0));
#pragma unroll
          for (int k = 0; k < NNET; ++k)
#pragma unroll
            for (int u = 0; u < 24; ++u) vmax[k] = fmaxf(vmax[k], fabsf(t[k][u]));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
      for (int o = 32; o > 0; o >>= 1) vmax[k] = fmaxf(vmax[k], __shfl_xor(vmax[k], o));
      if (lane == 0) red[4 * k + wave] = vmax[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
      const float vm = fmaxf(fmaxf(red[4 * k], red[4 * k + 1]), fmaxf(red[4 * k + 2], red[4 * k + 3]));
      int S = 0;
      if (vm > 0.f && vm < 3.0e38f) S = 14 - ilogbf(vm);
      S = S > 100 ? 100 : (S < -100 ? -100 : S);
      sc[k] = ldexpf(1.f, S); inv[k] = ldexpf(1.f, -S);
    }
  }

  __syncthreads();
#ifdef DW16_CLOCK
  const unsigned long long ce1 = __builtin_amdgcn_s_memrealtime();
#endif

  for (int unit = bx; unit < units; unit += gx) {
#ifdef DW16_CLOCK
    const unsigned long long cu0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (!primed) { unit_begin(unit); if (!DENSE) unit_requests(); }
    primed = false;
    const int b = ub;
    auto in_band = [&](int y) { return y >= q_lo && y < q_lo + rows; };
    const __amdgpu_buffer_rsrc_t rdense = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(DENSE ? a.dy_dense + (long)b * a.dy_dense_bstride : a.dy.dpool), 0, DENSE ? H * W * nout * 4 : 0, 0x00020000);
    if (DENSE) {
      for (int d = 0; d <= 2 * P; ++d) { dense_load(rdense, y0 + d); dense_store(d % G::RING_DY); }
      dense_load(rdense, y0 + 2 * P + 1);
      for (int d = P; d < P + 2; ++d)
        if (d - P < rows) { in_load(in_rs, y0 + d); in_store(d % G::RING_IN); }
      if (2 < rows) in_load(in_rs, q_lo + 2);
    } else {
    if (0 < rows) in_store(P % G::RING_IN);
    if (2 < rows) in_load(in_rs, q_lo + 2);
    dy_conv(0, in_band(y0));
    dy_store(0, 0); dy_store(1, 1);
    dy_conv(1, in_band(y0 + 2));
    dy_store(2, 0); dy_store(3, 1);
    if (1 < rows) in_store_from(sv2, (P + 1) % G::RING_IN);
    dy_conv(2, in_band(y0 + 4));
    dy_store(4, 0);                                   // position 5 (same cells) is stored by the first step
    }
    __syncthreads();
#ifdef DW16_CLOCK
    const unsigned long long cu1 = __builtin_amdgcn_s_memrealtime(); cpro += cu1 - cu0;
#endif

    for (int t0 = 0; t0 < rows + P; t0 += G::UNROLL) {
#pragma unroll
      for (int sq = 0; sq < G::UNROLL; ++sq) {
        const int t = t0 + sq;
        if (t < P) continue;                          // uniform
        if (t >= rows + P) break;
        {  // stage ahead: dY position t + P + 1 (its slot held position t - P - 1), input position t + 2
          const int d = t + P + 1, y = y0 + d;
          if (DENSE) {
            dense_store((sq + P + 1) % G::RING_DY);                       // requested one step ago
            dense_load(rdense, y + 1);
          } else {
            if ((d & 1) == 0) dy_conv(0, in_band(y));                     // requested one step ago
            dy_store((sq + P + 1) % G::RING_DY, (sq + P + 1) & 1);
            if ((d & 1) == 1) dy_issue((y + 1) >> 1, 0);                  // next pooled row, used from the next step on
          }
          if (t + 2 - P < rows) in_store((sq + 2) % G::RING_IN);
          if (t + 3 - P < rows) in_load(in_rs, y0 + t + 3);
        }
        // multiply input position t with dY positions t - P .. t + P (of every network: the A fragments are read once)
        const int islot = sq % G::RING_IN;
#pragma unroll
        for (int ch = 0; ch < NCHK; ++ch) {
          f16x8 bq[NNET][NCT][NPC];
#pragma unroll
          for (int k = 0; k < NNET; ++k)
#pragma unroll
            for (int j = 0; j < NCT; ++j)
#pragma unroll
              for (int pc = 0; pc < NPC; ++pc) {
                bq[k][j][pc] = lds_load<f16x8>(badr[j][sq], k * DYB + pc * G::DPC + ch * 64);
              }
          k16_u32x4 av[MTW];
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int off = islot * ROWB + ch * (2 * CP * 32) + mt * 32;
            const dw16_v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)off)));
            const dw16_v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)(off + 2 * CP))));
            const dw16_u32x2 u0 = __builtin_bit_cast(dw16_u32x2, r0), u1 = __builtin_bit_cast(dw16_u32x2, r1);
            av[mt] = (k16_u32x4){u0.x, u0.y, u1.x, u1.y};
          }
#pragma unroll
          for (int k = 0; k < NNET; ++k)
#pragma unroll
            for (int pc = NPC - 1; pc >= 0; --pc)
#pragma unroll
              for (int j = 0; j < NCT; ++j)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                  acc[k][j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av[mt]), bq[k][j][pc], acc[k][j][mt], 0, 0, 0);
        }
        __syncthreads();
      }
    }
  }

#ifdef DW16_CLOCK
  const unsigned long long ce2 = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- one partial per workgroup and network.  D tile mt holds rows m = 16 mt + 4 lj + r = CP kx + c', column n = (ky, o);
  // row c' = CIN of every kx is T: it goes through a wave-private LDS table, then dW = 2^-S (s_c G + t_c T)
  // (the epilogue scratch of network k reuses its own dY ring: texch [4 waves][KS][16], then the bias-gradient scratch [NCELL][256])
  const int nw = KS * G::KROW * nout;
  float* wsc = reinterpret_cast<float*>(dyring) + 4 * KS * 16 + CONV_THREADS * NCELL;           // [CIN] scale, [CIN] shift (ring 0, behind its scratch)
  // whitening scale / shift through LDS: read per (tile, row) below (global loads there were a chain of L2 round trips:
  // 8.4 us per workgroup, in-kernel probe)
  if (tid < CIN) { wsc[tid] = wsc_s; wsc[CIN + tid] = wsc_t; }
  __syncthreads();
#ifdef DW16_CLOCK
  const unsigned long long cq1 = __builtin_amdgcn_s_memrealtime();
#endif
  // (the table has one [KS][16] block per COLUMN TILE: a wave's own without SPLIT; with it the T row of a kx sits in the row tiles of ONE
  // of the two waves that share the column tiles, and both read it behind a workgroup barrier)
#pragma unroll
  for (int k = 0; k < NNET; ++k) {
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      float* tx = reinterpret_cast<float*>(dyring + k * DYB) + (NCT * wcol + j) * (KS * 16);
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const int m = CP * kx + CIN;                    // compile-time
        if (mt0 == ((m >> 4) / MTW) * MTW && lj == ((m & 15) >> 2)) tx[kx * 16 + li] = acc[k][j][(m >> 4) % MTW][m & 3];
      }
    }
  }
  if (G::SPLIT) __syncthreads();
  else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int k = 0; k < NNET; ++k) {
    float* part = batch.a[by + k].partial + (long)bx * batch.a[by + k].pstride;
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const float* tx = reinterpret_cast<const float*>(dyring + k * DYB) + (NCT * wcol + j) * (KS * 16);
      if (nvalid[j] && no[j] < nout) {
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * (mt0 + mt) + 4 * lj + r;
            const int kx = m / CP, c = m - kx * CP;
            if (kx < KS && c < CIN) {
              const float t = tx[kx * 16 + li];
              part[(nky[j] * G::KROW + kx * CIN + c) * nout + no[j]] = inv[k] * (wsc[c] * acc[k][j][mt][r] + wsc[CIN + c] * t);
            }
          }
        }
      }
    }
  }
  // bias gradient: per-thread cell sums -> LDS -> one thread per channel adds them in fixed order
#ifdef DW16_CLOCK
  const unsigned long long cq2 = __builtin_amdgcn_s_memrealtime();
#endif
  __syncthreads();
#ifdef DW16_CLOCK
  const unsigned long long cq3 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
  for (int k = 0; k < NNET; ++k) {
    float* dbs = reinterpret_cast<float*>(dyring + k * DYB) + 4 * KS * 16;
#pragma unroll
    for (int c = 0; c < NCELL; ++c) dbs[c * CONV_THREADS + tid] = cact[c] ? dbsum[k][c] : 0.f;
  }
  __syncthreads();
  if (tid < NNET * 64 && (tid & 63) < nout) {           // (one wave per network; channel o = the tasks T = o (mod nout), in task order)
    const int k = tid >> 6, o = tid & 63;
    const float* dbs = reinterpret_cast<const float*>(dyring + k * DYB) + 4 * KS * 16;
    float s = 0.f;
    const int ntask = nout * NCHK * 8;
    int T = o;
    for (; T + 3 * nout < ntask; T += 4 * nout) {         // (same order as a one-by-one loop; 8 LDS reads in flight)
      float t[8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int Tu = T + u * nout, tk = Tu / CONV_THREADS, th = Tu % CONV_THREADS;
        t[2 * u] = dbs[(2 * tk) * CONV_THREADS + th]; t[2 * u + 1] = dbs[(2 * tk + 1) * CONV_THREADS + th];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; T < ntask; T += nout) {
      const int tk = T / CONV_THREADS, th = T % CONV_THREADS;
      s += dbs[(2 * tk) * CONV_THREADS + th];
      s += dbs[(2 * tk + 1) * CONV_THREADS + th];
    }
    (batch.a[by + k].partial + (long)bx * batch.a[by + k].pstride)[nw + o] = s;
  }
#ifdef DW16_CLOCK
  if (tid == 0 && (bx % 211) == 7 && by == 0) printf("DW16CLK block %d: setup %llu, units (prologue %llu) %llu, epilogue %llu ticks (first barrier %llu, partial stores %llu, barrier %llu, bias %llu)\n", bx, ce1 - ce0, cpro, ce2 - ce1, __builtin_amdgcn_s_memrealtime() - ce2, cq1 - ce2, cq2 - cq1, cq3 - cq2, __builtin_amdgcn_s_memrealtime() - cq3);
#endif
}

template <int CIN, int KS, int NCHK, bool DENSE = false, int NPCS = F16_PIECES>
__global__ __launch_bounds__(CONV_THREADS, (Dw16Geom<CIN, KS, NCHK, NPCS>::MT > 8 ? 2 : DW16_WGS)) void conv_dw16_kernel(const ConvArgsN batch, int units_per_img, int band) {
  conv_dw16_body<CIN, KS, NCHK, DENSE, 1, NPCS>(batch, units_per_img, band, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}
// two networks that read the same images per workgroup (blockIdx.y selects the PAIR: networks 2y and 2y + 1)
template <int CIN, int KS, int NCHK, int NPCS = F16_PIECES>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_dw16_pair_kernel(const ConvArgsN batch, int units_per_img, int band) {
  conv_dw16_body<CIN, KS, NCHK, false, 2, NPCS>(batch, units_per_img, band, (int)blockIdx.x, 2 * (int)blockIdx.y, (int)gridDim.x);
}
// conv1 dW of the headline shape with the next minibatch's sample + statistics pass behind it in the same grid (conv1_dw_gather.hip)
int launch_conv1_dw_gather(cpp_ctx* ctx, const ConvArgsN& batch, int upi, int band, int grid, size_t lds_bytes, const GatherArgs& g, bool pair, bool exact);

// networks 2j and 2j + 1 of the batch read the same images with the same whitening: one workgroup can serve both
static inline bool conv_dw16_pairable(const ConvArgsN& batch) {
  if (batch.n < 2 || (batch.n & 1)) return false;
  for (int j = 0; j + 1 < batch.n; j += 2) {
    const ConvArgs &x = batch.a[j], &y = batch.a[j + 1];
    if (x.in != y.in || x.in_bstride != y.in_bstride || x.img_slot != y.img_slot || x.scale != y.scale || x.shift != y.shift ||
        x.dy_dense || y.dy_dense || x.pstride != y.pstride)
      return false;
  }
  return true;
}

template <int CIN, int KS, int NCHK, bool DENSE = false, bool PAIR = false, int NPCS = F16_PIECES>
static inline int conv_dw16_launch_t(cpp_ctx* ctx, const ConvArgsN& batch, int* grid_out) {
  typedef Dw16Geom<CIN, KS, NCHK, NPCS> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = PAIR ? (size_t)G::LDS_BYTES2 : (size_t)G::LDS_BYTES;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    if constexpr (PAIR) HIP_CHECK(hipFuncSetAttribute((const void*)conv_dw16_pair_kernel<CIN, KS, NCHK, NPCS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    else HIP_CHECK(hipFuncSetAttribute((const void*)conv_dw16_kernel<CIN, KS, NCHK, DENSE, NPCS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
#ifndef DW16_CAP
#define DW16_CAP 4
#endif
  // Units = (image, band of rows).  The bands are as short as ONE resident round of workgroups allows: a second, partial round runs
  // with the pipes half empty (cfg3 one-network: 1024 workgroups on 768 slots 81.3 us, 512 on 512 76.6 us; NAF's single trunk 51.8 ->
  // 47.1 us), fewer, longer units waste slots (9 channels fit four workgroups per CU: 1024 units 51.1 us, 512 units 61.0 us --
  // profiles/experiments/r03_dw16_cap_sweep.txt).  Slots per CU: what the runtime says for this instance and its LDS.
  static int wgs_per_cu[CPP_MAX_DEVICES] = {};
  if (wgs_per_cu[cpp_dev_slot(ctx)] == 0) {
    int nb = 0;
    hipError_t oe;
    if constexpr (PAIR) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_dw16_pair_kernel<CIN, KS, NCHK, NPCS>, CONV_THREADS, lds_bytes);
    else oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_dw16_kernel<CIN, KS, NCHK, DENSE, NPCS>, CONV_THREADS, lds_bytes);
    wgs_per_cu[cpp_dev_slot(ctx)] = (oe == hipSuccess && nb >= 1) ? (nb > DW16_CAP ? DW16_CAP : nb) : 2;
    (void)hipGetLastError();
  }
  const int launches = PAIR ? batch.n / 2 : batch.n;          // grid rows
  const int capacity = ctx->num_cus * wgs_per_cu[cpp_dev_slot(ctx)] / launches;   // <= num_cus * 4 / n: the partial buffers are sized for that
  int band = (a.H + 1) & ~1;
  while (band > 8 && (band / 2) % 2 == 0 && a.B * ((a.H + band / 2 - 1) / (band / 2)) <= capacity) band /= 2;
  const int upi = (a.H + band - 1) / band;
  const int units = a.B * upi;
  const int grid = units < capacity ? units : capacity;
  if (ctx->ride && !ctx->ride_done && ctx->ride_at_dw && ctx->ride_dtype == 1 && CIN == 18 && KS == 5 && NCHK == 2 && !DENSE) {
    ctx->ride_done = true;
    *grid_out = grid;
    return launch_conv1_dw_gather(ctx, batch, upi, band, grid, lds_bytes, *ctx->ride, PAIR, NPCS == F16_PIECES_EXACT);
  }
  if constexpr (PAIR) hipLaunchKernelGGL((conv_dw16_pair_kernel<CIN, KS, NCHK, NPCS>), dim3(grid, batch.n / 2), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band);
  else hipLaunchKernelGGL((conv_dw16_kernel<CIN, KS, NCHK, DENSE, NPCS>), dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band);
  LAUNCH_CHECK();
  *grid_out = grid;
  return 0;
}

// conv1 dW of f16 image batches with one whitening table (white_bstride == 0), pooled dY (no batch norm), 5x5, even CIN and W
int conv_dw16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool dense, const ConvArgsN& a, int* grid, bool* handled);
// the instances that serve two networks per workgroup (conv_dw16_pair.hip); *handled stays false for a geometry without one
int conv_dw16_pair_dispatch(cpp_ctx* ctx, int cin, int nchk, const ConvArgsN& a, int* grid, bool* handled);

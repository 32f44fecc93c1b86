// conv2's and conv3's dX on the bf16 matrix pipes, ROW-STREAMING with the weights resident in registers ("dxrs"): v_mfma_f32_16x16x32_bf16.
//
//   dX[b, y, x, c] = sum_{ky, kx, o} dZ[b, y - ky + P, x - kx + P, o] W[ky][kx][c][o]          (the gradient of base_network.py:111-123's convs)
//                  = sum_{ky', kx', o} dZ[b, y + ky' - P, x + kx' - P, o] W'[ky'][kx'][o][c],   W'[ky'][kx'][o][c] = W[KS-1-ky'][KS-1-kx'][c][o]
//
// i.e. a forward convolution of the unpooled gradient dZ with the flipped, transposed weights -- what conv_fwd_kyo_body<.., IN_DY> computes
// with v_mfma_f32_16x16x4_f32 (45 us alone at cfg3, two thirds of conv2's backward launch).  Here both f32 operands are split into three
// bf16 pieces (an f32 IS the sum of three bf16 numbers: conv_k16.h, B16 mode) and the six (CPP_PRECISION_EXACT: nine) largest piece
// products are issued: conv2 forward's arithmetic contract, in conv_rs16.h's formulation:
//
//   * the MFMA's 16 ROWS are the 10 input channels c of the layer (A operand = W', lane (li, lj) holds row li's 8 consecutive
//     k = (kx', o) of lane group lj); 5x5: KS x 2 chunks x 3 pieces = 30 A operands = 120 VGPRs (3x3: one chunk, 9 operands), built ONCE
//     per workgroup through LDS and resident;
//   * its 16 COLUMNS are pixels: a wave owns one 16-pixel tile of an image for all rows (16-wide rows: a workgroup = four images;
//     32-wide: two waves per image, two images; 64-wide: four waves, one image);
//   * dZ rows are REBUILT from the pooled gradient and the arg-max codes (as IN_DY does), split, and staged in the wave's own LDS slots
//     as three bf16 planes of 20 pixels (the tile + the SAME padding / the neighbours' two pixels) -- no barrier in the row loop; the
//     operand windows are 4-byte aligned LDS reads at per-lane addresses (20 bytes per pixel);
//   * the KS output rows a dZ row contributes to are KS accumulator sets (+ the one being written out), restarted through the C operand.
// 5x5: 60 MFMAs (90: nine products) per dZ row and wave; beside them ~40 VALU, 18 LDS accesses, one global load pair every second row and
// one 16-byte store per lane and row -- dealt out BETWEEN the MFMAs (the step below).  3x3: 18 MFMAs against the same staging work.
#pragma once
#include <type_traits>
#include "conv_k16.h"

typedef unsigned dxrs_u32x2 __attribute__((ext_vector_type(2)));

template <int KS_, int TPR>                                  // kernel size (5: conv2, 3: conv3); tiles per row: W = 16 TPR
struct DxRsGeom {
  static constexpr int KS = KS_, P = KS / 2, CH = KYO_NO, NCH = (KS * CH + 31) / 32, NSET = KS + 1;
  static constexpr int UNR = KS == 5 ? 6 : 12;                // steps per unrolled block: a multiple of NSET (sets), 2 (slots) and 6 (three pooled rows in flight)
  static constexpr int W = 16 * TPR, IPW = 4 / TPR;           // images per workgroup
  static constexpr int NW = KS * KS * CH * CH;                // the layer's weights
  static constexpr int AIMG_BYTES = KS * NCH * 3 * 1024;      // the A operands: one KB (64 lanes x 16 bytes) per (ky', chunk, piece)
  static constexpr int WPX = 16 + 2 * P;                      // pixels of a staged row: the tile and P on each side
  // a staged plane: WPX pixels x 10 channels bf16 + the over-read of the last chunk's windows (5x5: 400 -> 428 bytes; 3x3: 360 -> 364)
  static constexpr int PLB = ((2 * CH * 15 + 64 * (NCH - 1) + 64 + 15) & ~15) + 16;
  static constexpr int SLOTB = 3 * PLB;
  static constexpr int TRB = 640 + 16;                        // an output row of the tile on its way out (16 px x 10 ch f32) + a dump slot
  static constexpr int WVB = 2 * SLOTB + TRB;                 // per wave: two staged rows + the output row
  static constexpr int LDS_BYTES = AIMG_BYTES + 4 * WVB;
  static_assert((KS == 5 || KS == 3) && 4 % TPR == 0 && WVB % 16 == 0 && PLB >= 2 * CH * WPX + 8, "geometry");
};

// ORDER: B16_SIX / B16_NINE (the largest i + j of the piece products A_i B_j still issued: conv_k16.h)
template <int KSZ, int TPR, int ORDER>
__device__ __forceinline__ void conv_dx_rs_body(const ConvArgsN& batch, const int bx, const int by) {
  typedef DxRsGeom<KSZ, TPR> G;
  constexpr int KS = G::KS, P = G::P, CH = G::CH, NCH = G::NCH, NSET = G::NSET, UNR = G::UNR, W = G::W, Wp = W / 2;
  constexpr int PLB = G::PLB, SLOTB = G::SLOTB;
  const ConvArgs& a = batch.a[by];
  extern __shared__ __attribute__((aligned(16))) unsigned char dxrs_lds[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lj = lane >> 4;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int simg = swave / TPR, stile = swave % TPR;
  // a launch that would leave CUs without a workgroup (one network at 32-wide rows: 128) walks every image in TWO bands of output rows
  // [ylo, yhi); a band reads the P dZ rows above and below it and starts its walk at a multiple of NSET (the sets' and slots'
  // compile-time rotation), i.e. up to NSET - 1 rows early: those rows only reach output rows in front of the band, which are not stored
  const int nbands = a.nbands > 1 ? a.nbands : 1;
  const int band = bx % nbands;
  const int sb = (bx / nbands) * G::IPW + simg;
  const int H = a.H, Hp = H >> 1;
  const int ylo = nbands > 1 ? band * a.band_rows : 0, yhi = nbands > 1 ? min(H, ylo + a.band_rows) : H;
  const int qbeg = ((ylo > P ? ylo - P : 0) / UNR) * UNR;     // first dZ row walked
  const int qmf = min(H, yhi + P);                            // one past the last dZ row that reaches the band
  unsigned char* wvb = dxrs_lds + G::AIMG_BYTES + swave * G::WVB;
  // ---- dZ rows: lane l < 50 owns the channel pair (2 op, 2 op + 1) of pooled cell cw of the tile's window (pooled pixels 8 stile - 1 ..
  // 8 stile + 8: the tile and one cell of halo on each side); cells outside the image read zeros through the descriptors' range check
  constexpr unsigned BIG = 0x08000000u;
  const int cw = lane / 5, op = lane - cw * 5;
  const int ppx = stile * 8 - 1 + cw;
  const bool cell = lane < 50 && ppx >= 0 && ppx < Wp && sb < a.B;
  const __amdgpu_buffer_rsrc_t dp_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy.dpool + (long)(sb < a.B ? sb : 0) * a.dy.dpool_bstride), 0,
                                                                           Hp * Wp * CH * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t am_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.dy.amax + (long)(sb < a.B ? sb : 0) * Hp * Wp * CH), 0,
                                                                           Hp * Wp * CH, 0x00020000);
  const unsigned goff = cell ? (unsigned)((ppx * CH + 2 * op) * 4) : BIG;
  const unsigned coff = cell ? (unsigned)(ppx * CH + 2 * op) : BIG;
  // the cell's two pixels in the staged row (which starts at pixel 16 stile - P): 2 cw - 2 + P and the next one; 3x3: the first cell's even
  // and the last cell's odd pixel lie outside it.  (Those, and the lanes without a cell -- which hold zeros --, write into the first plane's tail.)
  constexpr int DUMP = PLB - 8;
  const int pe = 2 * cw - 2 + P, po = pe + 1;
  const uint32_t sadr_e = keep_in_vgpr(lds_addr(wvb + ((lane < 50 && pe >= 0 && pe < G::WPX) ? 2 * CH * pe + 4 * op : DUMP)));
  const uint32_t sadr_o = keep_in_vgpr(lds_addr(wvb + ((lane < 50 && po >= 0 && po < G::WPX) ? 2 * CH * po + 4 * op : DUMP)));
  // three pooled rows in flight (a block of NSET = 6 steps walks three of them: buffer = pooled row mod 3, a compile-time number): a
  // row is requested two steps before its conversion -- requested one step ahead, every second step opened with a wait for the round trip
  f32x2 rawg[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
  unsigned rawc[3] = {0u, 0u, 0u};
  unsigned pk[3] = {0u, 0u, 0u};                              // the pair's pieces: low half = channel 2 op
  unsigned mk[2][2] = {{0u, 0u}, {0u, 0u}};                   // [image-row parity ry][x parity]: which halves of pk belong to that pixel of the cell
  auto load_pooled = [&](const int buf, const int py) __attribute__((always_inline)) {      // (the row offset in the VGPR: the range check does not see soffset)
    const bool in = py < Hp;
    const dxrs_u32x2 g = __builtin_amdgcn_raw_buffer_load_b64(dp_rsrc, (int)(in ? goff + (unsigned)(py * (Wp * CH * 4)) : BIG), 0, 0);
    rawg[buf] = (f32x2){__uint_as_float(g.x), __uint_as_float(g.y)};
    rawc[buf] = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(am_rsrc, (int)(in ? coff + (unsigned)(py * (Wp * CH)) : BIG), 0, 0);
  };
  // (the first two pooled rows are on their way while the workgroup builds the operands)
  load_pooled(0, qbeg >> 1);
  load_pooled(1, (qbeg >> 1) + 1);
  unsigned char* aimg = dxrs_lds;                             // the A operands: [ky'][chunk][piece][lane][8 halves]
  // ---- the A operands W'[ky'][k = (kx', o)][c], three bf16 pieces each: built ONCE per workgroup (its waves serve one network), element
  // by element from coalesced loads -- a wave building its own 30 operands spent ~1800 instructions (3.7 us) on index arithmetic and splits
  {
    constexpr int NWT = (G::NW + CONV_THREADS - 1) / CONV_THREADS;
    float wr[NWT];
#pragma unroll
    for (int n = 0; n < NWT; ++n) { const int i = tid + n * CONV_THREADS; wr[n] = a.w[i < G::NW ? i : 0]; }
    for (int i = tid; i < G::AIMG_BYTES / 16; i += CONV_THREADS) reinterpret_cast<k16_u32x4*>(aimg)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
    // the wave's slots: zero (the planes' tails are read by chunk 1's windows, against zero weights)
    for (int i = lane; i < G::WVB / 16; i += 64) reinterpret_cast<k16_u32x4*>(wvb)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
      const int i = tid + n * CONV_THREADS;                   // W[ky][kx][c][o], o fastest
      if (i < G::NW) {
        const int o = i % CH, c = (i / CH) % CH, t = i / (CH * CH), kx = t % KS, ky = t / KS;
        const int k = (KS - 1 - kx) * CH + o, ch = k >> 5, lg = (k >> 3) & 3, e = k & 7;
        unsigned short h, m, l;
        k16_split3(a.wscale != 0.f ? wr[n] * a.wscale : wr[n], h, m, l);
        unsigned short* d = reinterpret_cast<unsigned short*>(aimg + (((KS - 1 - ky) * NCH + ch) * 3) * 1024 + (lg * 16 + c) * 16 + e * 2);
        d[0] = h; d[512] = m; d[1024] = l;
      }
    }
  }
  __syncthreads();
  if (sb >= a.B) return;                                      // (wave-uniform; no barrier below)
  k16_u32x4 wv[KS][NCH][3];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        wv[ky][ch][pc] = *reinterpret_cast<const k16_u32x4*>(aimg + ((ky * NCH + ch) * 3 + pc) * 1024 + lane * 16);

  auto convert = [&](const int buf) __attribute__((always_inline)) {
    const unsigned c0 = rawc[buf] & 0xFFu, c1 = rawc[buf] >> 8;
    const float g0 = (c0 & POOL_ACTIVE) ? rawg[buf][0] : 0.f, g1 = (c1 & POOL_ACTIVE) ? rawg[buf][1] : 0.f;
    unsigned short h0, m0, l0, h1, m1, l1;
    k16_split3(g0, h0, m0, l0); k16_split3(g1, h1, m1, l1);
    pk[0] = (unsigned)h0 | ((unsigned)h1 << 16); pk[1] = (unsigned)m0 | ((unsigned)m1 << 16); pk[2] = (unsigned)l0 | ((unsigned)l1 << 16);
    const unsigned p0 = c0 & 3u, p1 = c1 & 3u;                // window position 2 ry + rx of each channel's maximum
#pragma unroll
    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
      for (int rx = 0; rx < 2; ++rx)
        mk[ry][rx] = (p0 == (unsigned)(2 * ry + rx) ? 0xFFFFu : 0u) | (p1 == (unsigned)(2 * ry + rx) ? 0xFFFF0000u : 0u);
  };
  auto stage_row = [&](const int slot, const int ry) __attribute__((always_inline)) {      // dZ row 2 py + ry of the pooled row in pk / mk
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      lds_store(sadr_e, slot * SLOTB + p * PLB, pk[p] & mk[ry][0]);
      lds_store(sadr_o, slot * SLOTB + p * PLB, pk[p] & mk[ry][1]);
    }
  };
  // operand windows: pixel li of the tile, taps kx' = 0 .. 4 -> staged pixels li .. li + 4; 8 consecutive k = 16 bytes at 20 li + 64 ch + 16 lj
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

  // ---- outputs: accumulator layout (channels 4 lj .. 4 lj + 3 of pixel li) -> the row as it lies in memory, through LDS
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long)sb * a.out_bstride, 0, H * W * CH * 4, 0x00020000);
  unsigned char* trw = wvb + 2 * SLOTB;
  const uint32_t twA = keep_in_vgpr(lds_addr(trw + (lj < 3 ? (li * CH + 4 * lj) * 4 : 640)));
  const uint32_t twB = keep_in_vgpr(lds_addr(trw + (lj < 2 ? (li * CH + 4 * lj + 2) * 4 : 648)));
  const uint32_t trd = keep_in_vgpr(lds_addr(trw + (lane < 4 * CH ? 16 * lane : 0)));
  const unsigned eL = lane < 4 * CH ? (unsigned)(stile * 16 * CH * 4 + 16 * lane) : BIG;

  f32x4 acc[NSET];
#pragma unroll
  for (int s = 0; s < NSET; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  float vmx = 0.f;                                             // the largest |value| this lane stores (a.dx_imax)

  convert(0);
  stage_row(0, 0);
  read_x(0, 0);
  if (NCH > 1) read_x(NCH - 1, 0);
  __builtin_amdgcn_sched_barrier(0);

  auto step = [&](auto sqtag, auto gentag, const int q) __attribute__((always_inline)) {
    constexpr int SQ = decltype(sqtag)::value;
    constexpr bool GEN = decltype(gentag)::value;
    constexpr int SD = (SQ + 2 * NSET - P - 1) % NSET;       // the set of output row q - P - 1: complete since the last step
    constexpr int SLOT = SQ & 1;                             // the LDS slot of dZ row q (q0 is a multiple of NSET = 6)
    constexpr bool EVEN = (SQ & 1) == 0;
    const int yd = q - P - 1;
    auto epi_write = [&]() __attribute__((always_inline)) {
      lds_store(twA, 0, (f32x2){acc[SD][0], acc[SD][1]});
      lds_store(twB, 0, (f32x2){acc[SD][2], acc[SD][3]});
    };
    auto epi_store = [&]() __attribute__((always_inline)) {
      const f32x4 v = lds_load<f32x4>(trd, 0);
      const bool live = !GEN || (yd >= ylo && yd < yhi);
      if (KS == 5) {                                          // (conv2: conv1's dW scales its f16 pieces by this bound instead of scanning the rows)
        const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        vmx = fmaxf(vmx, live ? m : 0.f);
      }
      buffer_store_b128_held((k16_u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])},
                             out_rsrc, (int)(live ? eL : BIG), (live ? yd : 0) * (W * CH * 4));
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
              const bool restart = ch == 0 && sum == ORDER && pa == 2 && ky == 0;      // (the set output row q + 2 starts in)
              acc[s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k16_bf16x8, wv[ky][ch][pa]), __builtin_bit_cast(k16_bf16x8, xb[ch][pb]),
                                                               restart ? zero4 : acc[s], 0, 0, 0);
            }
          }
        }
    };
    constexpr int J = (SQ / 2) % 3;                          // pooled row q / 2 mod 3
    if (!GEN || q < qmf) {
      // One wave per SIMD (1024 tiles at cfg3): nobody else issues while this wave prepares the next row, so the preparation is dealt
      // out BETWEEN the MFMAs -- the matrix pipe takes a 16x16x32 every 16 cycles, the wave can issue two or three other instructions in
      // the shadow of each (as phases of their own the ~50 VALU + 17 LDS instructions of a row left the pipe idle for a third of the step).
      // Chunk 0's MFMAs carry dZ row q + 1 into LDS, the finished output row into its LDS slot and row q + 1's first operand windows out;
      // chunk 1's carry the conversion of the next pooled row (even steps), the output row's store and the second windows.
      constexpr int NMF = (ORDER == B16_NINE ? 9 : 6) * KS;  // MFMAs of a chunk
      if constexpr (NCH == 1) {
        // 3x3 (conv3): one chunk, 18 MFMAs a row -- the row is the staging work (the same as 5x5's) with the MFMAs in its shadow
        mfmas(std::integral_constant<int, 0>{});
        stage_row(SLOT ^ 1, EVEN ? 1 : 0);
        if (EVEN) load_pooled((J + 2) % 3, (q >> 1) + 2);
        epi_write();
        read_x(0, SLOT ^ 1);
        epi_store();
        if (EVEN) convert((J + 1) % 3);
        __builtin_amdgcn_sched_barrier(0);
      } else {
      mfmas(std::integral_constant<int, 0>{});
      stage_row(SLOT ^ 1, EVEN ? 1 : 0);                     // dZ row q + 1 (pooled row (q + 1) / 2: converted under the last even step)
      if (EVEN) load_pooled((J + 2) % 3, (q >> 1) + 2);      // (behind the last row: beyond the descriptors' range, zeros)
      epi_write();
      read_x(0, SLOT ^ 1);
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);               // the masked pieces, the load's addresses
        else if (i < 9) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // 3 ds_write2 of the row, 2 ds_write_b64 of the output row
        else if (i < 15) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);         // 6 ds_read2: the next operand windows
        if (EVEN && i == 4) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // the pooled row's two loads
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(std::integral_constant<int, 1>{});
      epi_store();
      if (EVEN) convert((J + 1) % 3);                        // pooled row q / 2 + 1, requested two steps ago
      read_x(1, SLOT ^ 1);
#pragma unroll
      for (int i = 0; i < NMF; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);              // the output row back from LDS
        else if (i == 4) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);         // ... and out
        else if (EVEN && i < NMF - 3) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        else if (i >= NMF - 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 6 ds_read2
      }
      __builtin_amdgcn_sched_barrier(0);
      }
    } else { epi_write(); epi_store(); }                     // (steps behind the image: the last output rows)
  };
  auto block = [&](auto gentag, const int q0) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(gentag)::value;
    const int qe = yhi + P + 1;                              // one past the last step: output row yhi - 1 leaves at step yhi + P
    if (GEN && q0 + 0 >= qe) return; step(std::integral_constant<int, 0>{}, gentag, q0 + 0);
    if (GEN && q0 + 1 >= qe) return; step(std::integral_constant<int, 1>{}, gentag, q0 + 1);
    if (GEN && q0 + 2 >= qe) return; step(std::integral_constant<int, 2>{}, gentag, q0 + 2);
    if (GEN && q0 + 3 >= qe) return; step(std::integral_constant<int, 3>{}, gentag, q0 + 3);
    if (GEN && q0 + 4 >= qe) return; step(std::integral_constant<int, 4>{}, gentag, q0 + 4);
    if (GEN && q0 + 5 >= qe) return; step(std::integral_constant<int, 5>{}, gentag, q0 + 5);
    if constexpr (UNR == 12) {
      if (GEN && q0 + 6 >= qe) return; step(std::integral_constant<int, 6>{}, gentag, q0 + 6);
      if (GEN && q0 + 7 >= qe) return; step(std::integral_constant<int, 7>{}, gentag, q0 + 7);
      if (GEN && q0 + 8 >= qe) return; step(std::integral_constant<int, 8>{}, gentag, q0 + 8);
      if (GEN && q0 + 9 >= qe) return; step(std::integral_constant<int, 9>{}, gentag, q0 + 9);
      if (GEN && q0 + 10 >= qe) return; step(std::integral_constant<int, 10>{}, gentag, q0 + 10);
      if (GEN && q0 + 11 >= qe) return; step(std::integral_constant<int, 11>{}, gentag, q0 + 11);
    }
  };
  int q0 = qbeg;
  block(std::true_type{}, q0); q0 += UNR;
  for (; q0 >= ylo + P + 1 && q0 + UNR <= qmf; q0 += UNR) block(std::false_type{}, q0);      // (every step multiplies and stores a row of the band)
  for (; q0 < yhi + P + 1; q0 += UNR) block(std::true_type{}, q0);
  if (KS == 5 && a.dx_imax) {                                 // the image's slots [4 tiles][2 bands]: a wave also fills those of the tiles and bands nobody owns
    for (int o = 32; o > 0; o >>= 1) vmx = fmaxf(vmx, __shfl_xor(vmx, o));
    if (lane == 0 && sb < a.B) {
      float* im = a.dx_imax + (long)sb * DX_IMAX_SLOTS;
      for (int t = stile; t < 4; t += TPR) {
        if (nbands > 1) im[2 * t + band] = vmx; else { im[2 * t] = vmx; im[2 * t + 1] = vmx; }
      }
    }
  }
}

template <int KSZ, int TPR, int ORDER>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_dx_rs_kernel(const ConvArgsN batch) {
  conv_dx_rs_body<KSZ, TPR, ORDER>(batch, blockIdx.x, blockIdx.y);
}

// conv2's (5x5; 32- and 64-wide rows) and conv3's (3x3; 16-, 32- and 64-wide) dX: IN_DY (pooled gradient + codes in, plain rows out), 10 -> 10 channels
int conv_dx_rs_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled);
bool conv_dx_rs_ok(const cpp_ctx* ctx, int cin, int ks, int H, int W, int nout);

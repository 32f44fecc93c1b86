// conv2 weight / bias gradient on the bf16 matrix pipes with f32-exact operands ("dwb16"): v_mfma_f32_16x16x32_bf16.
//
//   dW[ky][kx][c][o] = sum_{b,q,x} in[b, q, x + kx - P, c] * dY[b, q - ky + P, x, o]
//
// Both operands are f32 here (pooled activations of conv1, routed pooled gradient).  An f32 is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits, bf16 has f32's exponent range, so no scaling is needed), every bf16 x bf16
// product is exact in the f32 accumulator, and all 3 x 3 = 9 products of the pieces are issued: the f32-MFMA kernel's
// arithmetic (conv_dw_kyo.h: exact products, f32 accumulation) at 9 x 16 cycles per 32 pixels instead of 8 x 32.
// (Dropping the small cross terms would NOT be exact and is not done.)
//
// Structure of conv_dw16.h: per input row q,  D[m = (kx,c), n = (ky,o)] += A[m, pixel] * B[pixel, n]; wave w owns column tile w
// of (ky,o) and all MT row tiles; units = (image, band of rows); one partial per workgroup.  The input row is split into
// its three pieces while it is staged: three planes in LDS at a pixel pitch of CP halves (8-byte aligned pixels), read
// through ds_read_b64_tr_b16 like the f16 kernel's single plane; dY pieces sit in LDS as [piece][o][chunk][g][8 halves].
#pragma once
#include "conv_dw16.h"

typedef __bf16 dwb_bf16x8 __attribute__((ext_vector_type(8)));

// round-to-nearest-even bf16 of a finite f32, as the 16 high bits
__device__ __forceinline__ unsigned dwb_bf16_bits(float x) {
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = h + m + l exactly (three bf16): returns the three 16-bit patterns
__device__ __forceinline__ void dwb_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned hb = dwb_bf16_bits(x);
  const float r1 = x - __uint_as_float(hb << 16);
  const unsigned mb = dwb_bf16_bits(r1);
  const float r2 = r1 - __uint_as_float(mb << 16);
  h = (unsigned short)hb; m = (unsigned short)mb; l = (unsigned short)(__float_as_uint(r2) >> 16);     // r2 has <= 8 significant bits
}

template <int CIN, int KS, int NCHK>
struct Dwb16Geom {
  static constexpr int P = KS / 2, NO = KYO_NO, NPC = 3, NPA = 3;
  static constexpr int CP0 = (CIN + 3) & ~3;
  static constexpr int CP = ((CP0 / 2) % 8 == 0) ? CP0 + 4 : CP0;       // channel pitch of a pixel in LDS (halves), see conv_dw16.h
  static constexpr int MROWS = KS * CP, MT = (MROWS + 15) / 16;
  static constexpr int KROW = KS * CIN;
  static constexpr int WPAD = 32 * NCHK;
  static constexpr int ROWH = ((CP * (WPAD + KS - 1) + 16 * MT - MROWS) + 7) & ~7;     // halves per staged row plane (+ m over-read)
  static constexpr int ROWB = ROWH * 2;
  static constexpr int DOST = 64 * NCHK + 32;
  static constexpr int DPC = NO * DOST;
  static constexpr int DSLOT = ((NPC * DPC - 192 + 255) / 256) * 256 + 192;
  static constexpr int RING_IN = 3, RING_DY = 6, UNROLL = 6;
  static constexpr int NCELL = (16 * NCHK * NO + CONV_THREADS - 1) / CONV_THREADS;
  static constexpr int NVIN = (WPAD * CIN + CONV_THREADS - 1) / CONV_THREADS;          // f32 elements of an input row per thread
  static constexpr int IN_BYTES = RING_IN * NPA * ROWB, DY_BYTES = RING_DY * DSLOT;
  static constexpr int LDS_BYTES = ((IN_BYTES + 15) & ~15) + DY_BYTES + 64;
  static_assert(DY_BYTES >= CONV_THREADS * NCELL * 4, "epilogue scratch fits the dY ring");
};

// (bx, by, gx): the workgroup's place in a (gx, networks) grid (its own launch, or a slice of a shared one: conv2_bwd_pair.hip)
template <int CIN, int KS, int NCHK, int ORDER = B16_SIX>
__device__ __forceinline__ void conv_dwb16_body(const ConvArgsN& batch, int units_per_img, int band, const int bx, const int by, const int gx) {
  typedef Dwb16Geom<CIN, KS, NCHK> G;
  constexpr int P = G::P, NO = G::NO, MT = G::MT, CP = G::CP, NPC = G::NPC, NPA = G::NPA, ROWB = G::ROWB, DSLOT = G::DSLOT;
  constexpr int SLOTB = NPA * ROWB;                  // bytes per input ring slot (three planes)
  static_assert((KS * NO + 15) / 16 == 4, "one column tile per wave");
  const ConvArgs& a = batch.a[by];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* inring = lds_raw;                                         // [3 slots][3 planes][ROWB]
  unsigned char* dyring = lds_raw + ((G::IN_BYTES + 15) & ~15);            // [6][DSLOT]
  float* dbs = reinterpret_cast<float*>(dyring);                           // epilogue: bias-gradient scratch [NCELL][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;
  const int units = a.B * units_per_img;

  for (int i = tid; i < (int)(((G::IN_BYTES + 15) & ~15) + G::DY_BYTES) / 16; i += CONV_THREADS)
    reinterpret_cast<float4*>(lds_raw)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- input row staging: f32 elements (x, c) of the row -> three bf16 planes at pixel pitch CP
  float sv[G::NVIN];
  bool sact[G::NVIN];
  uint32_t sdst[G::NVIN];
#pragma unroll
  for (int i = 0; i < G::NVIN; ++i) {
    const int d = tid + CONV_THREADS * i;
    const int x = d / CIN, c = d - x * CIN;
    sact[i] = d < W * CIN;
    sdst[i] = keep_in_vgpr(lds_addr(inring + 2 * (CP * (x + P) + c)));
    sv[i] = 0.f;
  }
  const int rowbytes = W * CIN * 4;
  float sv2[G::NVIN];                                 // (the unit prologue has two rows in flight: one L2 round trip instead of two)
  auto in_load_to = [&](float (&dst)[G::NVIN], const __amdgpu_buffer_rsrc_t& rs, int q) {
#pragma unroll
    for (int i = 0; i < G::NVIN; ++i)
      if (sact[i]) dst[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (tid + CONV_THREADS * i) * 4, q * rowbytes, 0));
  };
  auto in_load = [&](const __amdgpu_buffer_rsrc_t& rs, int q) { in_load_to(sv, rs, q); };
  auto in_store_from = [&](const float (&src)[G::NVIN], int slot) {
#pragma unroll
    for (int i = 0; i < G::NVIN; ++i)
      if (sact[i]) {
        unsigned short h, m, l;
        dwb_split3(src[i], h, m, l);
        lds_store(sdst[i], slot * SLOTB, h);
        lds_store(sdst[i], slot * SLOTB + ROWB, m);
        lds_store(sdst[i], slot * SLOTB + 2 * ROWB, l);
      }
  };
  auto in_store = [&](int slot) { in_store_from(sv, slot); };

  // ---- dY staging (as conv_dw16.h; bf16 pieces, no scaling)
  constexpr int NCELL = G::NCELL;
  bool cact[NCELL];
  uint32_t cdst[NCELL];
  float cg[NCELL], dbsum[NCELL];
  unsigned short cpc[NCELL][NPC];
  int ccode[NCELL];
#pragma unroll
  for (int c = 0; c < NCELL; ++c) {
    const int idx = tid + CONV_THREADS * c;
    const int px = idx / nout, o = idx - px * nout;
    cact[c] = idx < Wp * nout;
    const int x = 2 * px, ch = x >> 5, w = x & 31;
    const int g = 2 * ((w >> 1) & 1) + (w >> 4), e = (w >> 2) & 3;
    cdst[c] = keep_in_vgpr(lds_addr(dyring + o * G::DOST + ch * 64 + g * 16 + e * 2));
    cg[c] = 0.f; ccode[c] = 0; dbsum[c] = 0.f;
#pragma unroll
    for (int pc = 0; pc < NPC; ++pc) cpc[c][pc] = 0;
  }
  float rdv[3][NCELL];
  int rcd[3][NCELL];
  auto dy_issue = [&](const __amdgpu_buffer_rsrc_t& rp, const __amdgpu_buffer_rsrc_t& rd,
                      const __amdgpu_buffer_rsrc_t& rc, int py, int set) {
    const bool rowok = py >= 0 && py < Hp;           // uniform
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      rdv[set][c] = 0.f; rcd[set][c] = 0;
      if (rowok && cact[c]) {
        const int vo = (tid + CONV_THREADS * c) * 4, so = py * Wp * nout * 4;
        rdv[set][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, vo, so, 0));
        rcd[set][c] = __builtin_amdgcn_raw_buffer_load_b8(rc, vo >> 2, so >> 2, 0);
      }
    }
  };
  auto dy_conv = [&](int set, bool count) {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      cg[c] = (rcd[set][c] & 4) ? rdv[set][c] : 0.f;      // bit 2 of the code byte: the pooled output was > 0
      ccode[c] = rcd[set][c] & 3;
      if (count) dbsum[c] += cg[c];
      dwb_split3(cg[c], cpc[c][0], cpc[c][1], cpc[c][2]);
    }
  };
  auto dy_store = [&](int slot, int ry) {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      if (cact[c]) {
        const bool s0 = ccode[c] == 2 * ry, s1 = ccode[c] == 2 * ry + 1;
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
          lds_store(cdst[c], slot * DSLOT + pc * G::DPC, (unsigned short)(s0 ? cpc[c][pc] : 0));
          lds_store(cdst[c], slot * DSLOT + pc * G::DPC + 8, (unsigned short)(s1 ? cpc[c][pc] : 0));
        }
      }
    }
  };

  // ---- MFMA operands (pixel dealing and transpose reads as conv_dw16.h)
  const int tj = (lane >> 2) & 3, tq = lane & 3;
  const uint32_t aadr = keep_in_vgpr(lds_addr(inring + 2 * (CP * (16 * (lj & 1) + 4 * tj + 2 * (lj >> 1)) + 4 * tq)));
  const int n = 16 * wave + li;
  const bool nvalid = n < KS * NO;
  const int nky = nvalid ? n / NO : 0, no = nvalid ? n % NO : 0;
  uint32_t badr[G::UNROLL];
#pragma unroll
  for (int sq = 0; sq < G::UNROLL; ++sq) {
    const int slot = (sq - nky + P + G::RING_DY) % G::RING_DY;
    badr[sq] = keep_in_vgpr(lds_addr(dyring + slot * DSLOT + no * G::DOST + lj * 16));
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  __syncthreads();

  for (int unit = bx; unit < units; unit += gx) {
    const int b = unit / units_per_img;
    const int q_lo = (unit - b * units_per_img) * band;
    const int rows = min(band, H - q_lo);            // band and q_lo are even
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>((const float*)a.in + (long)b * a.in_bstride), 0, H * rowbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy.pool + (long)b * a.dy.pool_bstride), 0, Hp * Wp * nout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy.dpool + (long)b * a.dy.dpool_bstride), 0, Hp * Wp * nout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.dy.amax + (long)b * Hp * Wp * nout), 0, Hp * Wp * nout, 0x00020000);
    const int y0 = q_lo - P;
    auto in_band = [&](int y) { return y >= q_lo && y < q_lo + rows; };
    if (0 < rows) in_load(in_rs, q_lo);
    if (1 < rows) in_load_to(sv2, in_rs, q_lo + 1);
    dy_issue(rp, rd, rc, y0 >> 1, 0);
    dy_issue(rp, rd, rc, (y0 >> 1) + 1, 1);
    dy_issue(rp, rd, rc, (y0 >> 1) + 2, 2);
    if (0 < rows) in_store(P % G::RING_IN);
    if (2 < rows) in_load(in_rs, q_lo + 2);
    dy_conv(0, in_band(y0));
    dy_store(0, 0); dy_store(1, 1);
    dy_conv(1, in_band(y0 + 2));
    dy_store(2, 0); dy_store(3, 1);
    if (1 < rows) in_store_from(sv2, (P + 1) % G::RING_IN);
    dy_conv(2, in_band(y0 + 4));
    dy_store(4, 0);
    __syncthreads();

    for (int t0 = 0; t0 < rows + P; t0 += G::UNROLL) {
#pragma unroll
      for (int sq = 0; sq < G::UNROLL; ++sq) {
        const int t = t0 + sq;
        if (t < P) continue;                          // uniform
        if (t >= rows + P) break;
        {
          const int d = t + P + 1, y = y0 + d;
          if ((d & 1) == 0) dy_conv(0, in_band(y));
          dy_store((sq + P + 1) % G::RING_DY, (sq + P + 1) & 1);
          if ((d & 1) == 1) dy_issue(rp, rd, rc, (y + 1) >> 1, 0);
          if (t + 2 - P < rows) in_store((sq + 2) % G::RING_IN);
          if (t + 3 - P < rows) in_load(in_rs, y0 + t + 3);
        }
        const int islot = sq % G::RING_IN;
#pragma unroll
        for (int ch = 0; ch < NCHK; ++ch) {
          dwb_bf16x8 bq[NPC];
#pragma unroll
          for (int pc = 0; pc < NPC; ++pc) bq[pc] = __builtin_bit_cast(dwb_bf16x8, lds_load<k16_u32x4>(badr[sq], pc * G::DPC + ch * 64));
#pragma unroll
          for (int pa = NPA - 1; pa >= 0; --pa) {       // small pieces first
            k16_u32x4 av[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const int off = islot * SLOTB + pa * ROWB + ch * (2 * CP * 32) + mt * 32;
              const dw16_v4s r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)off)));
              const dw16_v4s r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  reinterpret_cast<__attribute__((address_space(3))) dw16_v4s*>((uintptr_t)(aadr + (uint32_t)(off + 2 * CP))));
              const dw16_u32x2 u0 = __builtin_bit_cast(dw16_u32x2, r0), u1 = __builtin_bit_cast(dw16_u32x2, r1);
              av[mt] = (k16_u32x4){u0.x, u0.y, u1.x, u1.y};
            }
#pragma unroll
            for (int pc = NPC - 1; pc >= 0; --pc) {
              if (pa + pc > ORDER) continue;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dwb_bf16x8, av[mt]), bq[pc], acc[mt], 0, 0, 0);
            }
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- one partial per workgroup: D tile mt holds rows m = 16 mt + 4 lj + r = CP kx + c, column n = (ky, o)
  float* part = a.partial + (long)bx * a.pstride;
  const int nw = KS * G::KROW * nout;
  if (nvalid && no < nout) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * mt + 4 * lj + r;
        const int kx = m / CP, c = m - kx * CP;
        if (kx < KS && c < CIN) part[(nky * G::KROW + kx * CIN + c) * nout + no] = acc[mt][r];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCELL; ++c) dbs[c * CONV_THREADS + tid] = cact[c] ? dbsum[c] : 0.f;
  __syncthreads();
  if (tid < nout) {
    float s = 0.f;
    const int nidx = Wp * nout;
    int idx = tid;
    for (; idx + 7 * nout < nidx; idx += 8 * nout) {      // (same order as a one-by-one loop, 8 LDS reads in flight)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int i = idx + u * nout; t[u] = dbs[(i / CONV_THREADS) * CONV_THREADS + (i % CONV_THREADS)]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; idx < nidx; idx += nout) s += dbs[(idx / CONV_THREADS) * CONV_THREADS + (idx % CONV_THREADS)];
    part[nw + tid] = s;
  }
}

template <int CIN, int KS, int NCHK, int ORDER = B16_SIX>
__global__ __launch_bounds__(CONV_THREADS, 3) void conv_dwb16_kernel(const ConvArgsN batch, int units_per_img, int band) {
  conv_dwb16_body<CIN, KS, NCHK, ORDER>(batch, units_per_img, band, blockIdx.x, blockIdx.y, gridDim.x);
}

#ifndef DWB16_CAP
#define DWB16_CAP 4
#endif
template <int CIN, int KS, int NCHK, int ORDER = B16_SIX>
static inline int conv_dwb16_launch_t(cpp_ctx* ctx, const ConvArgsN& batch, int* grid_out) {
  typedef Dwb16Geom<CIN, KS, NCHK> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)G::LDS_BYTES;
  auto kern = conv_dwb16_kernel<CIN, KS, NCHK, ORDER>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  static const int cap = cpp_switch_int("CPP_DWB16_CAP", DWB16_CAP);      // resident workgroups per CU the bands are chosen for
  const int capacity = ctx->num_cus * cap / batch.n;   // <= conv_dw_kyo_grid (4 per CU): the partial buffers are sized for that
  int band = (a.H + 1) & ~1;
  while (band > 8 && (band / 2) % 2 == 0 && a.B * ((a.H + band / 2 - 1) / (band / 2)) <= capacity) band /= 2;
  const int upi = (a.H + band - 1) / band;
  const int units = a.B * upi;
  const int grid = units < capacity ? units : capacity;
  if (ctx->pair && ctx->pair->layer == 1 && CIN == 10 && KS == 5 && NCHK == 1) {      // leaves with conv2's dX (conv2_bwd_pair.hip)
    ctx->pair->dw = batch; ctx->pair->dw_gx = grid; ctx->pair->dw_lds = lds_bytes; ctx->pair->have_dw = true;
    ctx->pair->upi = upi; ctx->pair->band = band;
    *grid_out = grid;
    return 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band);
  LAUNCH_CHECK();
  *grid_out = grid;
  return 0;
}

// conv2 dW (f32 pooled activations in, pooled dY, 5x5, 10 channels, even W <= 64 and H)
int conv_dwb16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, int* grid, bool* handled);

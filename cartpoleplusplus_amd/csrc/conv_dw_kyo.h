// Weight / bias gradient of a 5x5 "SAME" convolution followed by ReLU + 2x2 max-pool, "(ky, o)-column" formulation
// on v_mfma_f32_16x16x4_f32 (exact f32; base_network.py:103-123 backward).
//
//   dW[ky][(kx,c)][o] = sum_{b, q, x} in[b, q, x + kx - P, c] * dY[b, q - ky + P, x, o]
//
// For ONE input row q the MFMA reduces over the row's pixels (K = 4 pixels per instruction):
//   D[m = (kx,c), n = (ky,o)] += A[m, pixel] * B[pixel, n],   A = staged input row q,  B = the KS dY rows around q
// so the 10 filters share the MFMA's 16 columns with the vertical taps (50 of 64 columns useful; conv_dw_kernel keeps
// one accumulator tile per (ky, 16 (kx,c)) and uses 10 of 16).  conv1: 24 instead of 30 accumulator tiles.
//
// Workgroup = 4 waves = the 4 column tiles: wave w owns columns 16w..16w+15 of (ky,o) and ALL MT row tiles, i.e.
// only MT*4 accumulator registers and no cross-wave reduction; every wave walks all pixels of the row.  Rows of m are
// dealt to lanes as m = MT*i + mt, so a lane's A operands of one pixel are MT contiguous floats (ds_read_b64 pairs);
// pixels go to (step s, lane group k) as x = 4s + k, and dY rows sit in LDS as [o][k][s], so one ds_read_b128 holds a
// lane's B operands of 4 consecutive steps.
//
// A unit of work is (image, band of rows); a persistent workgroup takes units g, g+G, ... in order and keeps the sums
// in registers until the end (one partial per workgroup, reduced in fixed order by conv_dw_reduce_kernel).  Within a
// unit, position t of the band's row stream carries input row q = q_lo - P + t and dY row y = q_lo - P + t; the step at
// t multiplies input position t with dY positions t-P .. t+P.  dY rows live in a ring of 6 (5 live + the one being
// written), input rows in a ring of 3; the step loop is unrolled by 6 so that every ring slot -- and with it every LDS
// address -- is a compile-time offset from registers set up once.  f32-MFMA time is VALU time on this chip, so the
// loop carries no vector address arithmetic at all.
#pragma once
#include "conv_kyo.h"

template <int CIN, int KS, int NS_>
struct DwKyoGeom {
  static constexpr int P = KS / 2, NO = KYO_NO;
  static constexpr int NT = (KS * NO + 15) / 16;             // column tiles == waves
  static constexpr int KROW = KS * CIN, MT = (KROW + 15) / 16;
  static constexpr int NS = NS_, WPAD = 4 * NS;               // k-steps per row (multiple of 4), padded row width
  static constexpr int FP = (4 - (P * CIN) % 4) % 4;
  static constexpr int ROWF = ((FP + (WPAD + KS - 1) * CIN + 16 * MT) + 3) & ~3;   // staged input row (+ m over-read)
  static constexpr int KST = NS + 4, OST = 4 * KST + 4;      // dY row [o][k][s] with skewed strides: the staging writes of one
                                                             // wave (cells (px, o), o fastest) and the B reads spread over the banks
  static constexpr int DROW = NO * OST + 8;                  // (+ skew between ring slots)
  static constexpr int RING_IN = 3, RING_DY = 6, UNROLL = 6;
  static constexpr int WHF = 2 * ((CIN + 8 + 3) & ~3);
  static constexpr int NCELL = (2 * NS * NO + CONV_THREADS - 1) / CONV_THREADS;    // pooled cells of a row per thread
  static constexpr int LDS_FLOATS = RING_IN * ROWF + RING_DY * DROW + WHF + CONV_THREADS * NCELL;
};

// CHB: bytes per input staging chunk (16 / 8 / 4, as in conv_fwd_kyo_kernel).  DENSE: dY comes as dense f32 rows
// (a.dy_dense: batch norm's dz) instead of being rebuilt from the pooled gradient; that mode also takes odd heights and
// the 3x3 layer (two column tiles: waves 2 and 3 then only help with the staging).
template <int CIN, int KS, int NS_, int IN_MODE, int CHB = 16, bool DENSE = false>
__global__ __launch_bounds__(CONV_THREADS, 4) void conv_dw_kyo_kernel(const ConvArgsN batch, int units_per_img,
                                                                    int band) {
  typedef DwKyoGeom<CIN, KS, NS_> G;
  typedef typename StageType<IN_MODE>::type ST;
  static_assert(G::NT <= 4 && (DENSE || G::NT == 4), "one column tile per wave");
  constexpr bool WHITEN = (IN_MODE == IN_F16_WHITEN || IN_MODE == IN_F32_WHITEN);
  constexpr int P = G::P, NO = G::NO, MT = G::MT, NS = G::NS, ROWF = G::ROWF, DROW = G::DROW;
  constexpr int EPC = CHB / (int)sizeof(ST);
  constexpr bool A64 = (CIN % 2 == 0) && (MT % 2 == 0) && (G::FP % 2 == 0);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const ConvArgs& a = batch.a[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* inring = lds;                               // [3][ROWF]
  float* dyring = lds + G::RING_IN * ROWF;           // [6][DROW]
  float* whs = dyring + G::RING_DY * DROW;           // whitening scale[c], c < CIN + 8 (wrap-around); then shift[]
  float* wht = whs + G::WHF / 2;
  float* dbs = whs + G::WHF;                         // bias-gradient scratch [NCELL][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;

  for (int i = tid; i < (G::RING_IN * ROWF + G::RING_DY * DROW) / 4; i += CONV_THREADS)
    reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (WHITEN) {
    for (int c = tid; c < CIN + 8; c += CONV_THREADS) { whs[c] = a.scale[c % CIN]; wht[c] = a.shift[c % CIN]; }
  }

  // ---- input row staging (as in conv_fwd_kyo_kernel): fixed chunk per thread, buffer addressing
  const int cpr = (W * CIN) / EPC;
  constexpr int NVMAX = (G::WPAD * CIN / EPC + CONV_THREADS - 1) / CONV_THREADS;
  uint4 sv[NVMAX];
  unsigned sbyte[NVMAX];
  bool sact[NVMAX];
  uint32_t sdst[NVMAX], swh[NVMAX];
#pragma unroll
  for (int i = 0; i < NVMAX; ++i) {
    const int j = tid + CONV_THREADS * i;
    sact[i] = j < cpr;
    sbyte[i] = (unsigned)(j * EPC) * (unsigned)sizeof(ST);
    sdst[i] = keep_in_vgpr(lds_addr(inring + G::FP + P * CIN + j * EPC));
    swh[i] = keep_in_vgpr(lds_addr(whs + (j * EPC) % CIN));
  }
  const int rowbytes = W * CIN * (int)sizeof(ST);
  auto in_load = [&](const __amdgpu_buffer_rsrc_t& rs, int q) {
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      if (sact[i]) {
        if (CHB == 16) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)sbyte[i], q * rowbytes, 0);
          sv[i] = make_uint4(v.x, v.y, v.z, v.w);
        } else if (CHB == 8) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)sbyte[i], q * rowbytes, 0);
          sv[i] = make_uint4(v.x, v.y, 0u, 0u);
        } else {
          sv[i] = make_uint4(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)sbyte[i], q * rowbytes, 0), 0u, 0u, 0u);
        }
      }
    }
  };
  auto in_store = [&](int slot) {
#pragma unroll
    for (int i = 0; i < NVMAX; ++i) {
      if (sact[i]) {
        float x[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) x[k] = ChunkOps<ST>::get(sv[i], k);
        if (WHITEN) {
#pragma unroll
          for (int k = 0; k < EPC; k += 2) {
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
        if (EPC >= 4) {
#pragma unroll
          for (int k = 0; k + 3 < EPC; k += 4)
            lds_store(sdst[i], slot * ROWF * 4 + 4 * k, (f32x4){x[k], x[k + 1], x[k + 2], x[k + 3]});
        } else if (EPC == 2) {
          lds_store(sdst[i], slot * ROWF * 4, (f32x2){x[0], x[1]});
        } else {
          lds_store(sdst[i], slot * ROWF * 4, x[0]);
        }
      }
    }
  };

  // ---- dY staging: a thread owns pooled cells idx = px * nout + o of a pooled row (the same cells for every row):
  // one (gm, code) pair serves the two image rows of the pooled row.  LDS row layout [o][k = x % 4][s = x / 4].
  constexpr int NCELL = G::NCELL;
  bool cact[NCELL];
  uint32_t cdst[NCELL];
  float cg[NCELL], dbsum[NCELL];
  int ccode[NCELL];
#pragma unroll
  for (int c = 0; c < NCELL; ++c) {
    const int idx = tid + CONV_THREADS * c;
    const int px = idx / nout, o = idx - px * nout;
    cact[c] = idx < Wp * nout;
    cdst[c] = keep_in_vgpr(lds_addr(dyring + o * G::OST + ((2 * px) & 3) * G::KST + (px >> 1)));
    cg[c] = 0.f; ccode[c] = 0; dbsum[c] = 0.f;
  }
  // request the cells of pooled row py (zero outside the image) into raw register set `set`; dy_conv turns them into
  // (masked gradient, code) one step later, so the global latency hides under a step's MFMAs.  Each cell is counted
  // once for the bias gradient.
  float rdv[2][NCELL];
  int rcd[2][NCELL];
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
    }
  };
  // write image row (parity ry of its pooled row) into ring slot `slot`
  auto dy_store = [&](int slot, int ry) {
#pragma unroll
    for (int c = 0; c < NCELL; ++c) {
      if (cact[c]) {
        const float v0 = ccode[c] == 2 * ry ? cg[c] : 0.f, v1 = ccode[c] == 2 * ry + 1 ? cg[c] : 0.f;
        lds_store(cdst[c], slot * DROW * 4, v0);
        lds_store(cdst[c], slot * DROW * 4 + G::KST * 4, v1);
      }
    }
  };

  // ---- dense dY rows (DENSE): 8-byte chunks = channels (o0, o0 + 1) of one pixel; same LDS layout [o][k][s]
  constexpr int NDCH = DENSE ? (G::WPAD * NO / 2 + CONV_THREADS - 1) / CONV_THREADS : 1;
  bool dact[NDCH]; uint32_t ddst[NDCH]; f32x2 dreg[NDCH];
#pragma unroll
  for (int c = 0; c < NDCH; ++c) {
    const int j = tid + CONV_THREADS * c;              // chunk: flattened (x, o) elements 2j, 2j + 1
    const int x = (2 * j) / nout, o = (2 * j) - x * nout;
    dact[c] = DENSE && 2 * j < W * nout && (nout % 2) == 0;
    ddst[c] = keep_in_vgpr(lds_addr(dyring + o * G::OST + (x & 3) * G::KST + (x >> 2)));
    dreg[c] = (f32x2){0.f, 0.f};
  }
  auto dense_load = [&](const __amdgpu_buffer_rsrc_t& rs, int y) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const bool rowok = y >= 0 && y < H;               // uniform
#pragma unroll
    for (int c = 0; c < NDCH; ++c) {
      dreg[c] = (f32x2){0.f, 0.f};
      if (rowok && dact[c]) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (tid + CONV_THREADS * c) * 8, y * W * nout * 4, 0);
        dreg[c] = (f32x2){__uint_as_float(v.x), __uint_as_float(v.y)};
      }
    }
  };
  auto dense_store = [&](int slot) {
#pragma unroll
    for (int c = 0; c < NDCH; ++c) {
      if (dact[c]) {
        lds_store(ddst[c], slot * DROW * 4, dreg[c].x);
        lds_store(ddst[c], slot * DROW * 4 + G::OST * 4, dreg[c].y);
      }
    }
  };

  // ---- MFMA operands: A = input, lane (i = li, k = lj) reads m = MT*i .. MT*i + MT-1 of pixel x = 4 s + k;
  //                     B = dY,    lane (k = lj, j = li) reads column n = 16 wave + j = (ky, o) of the same pixels
  const uint32_t aadr = keep_in_vgpr(lds_addr(inring + G::FP + lj * CIN + MT * li));
  const int n = 16 * wave + li;
  const bool nvalid = n < KS * NO;
  const int nky = nvalid ? n / NO : 0, no = nvalid ? n % NO : 0;
  uint32_t badr[G::UNROLL];
#pragma unroll
  for (int sq = 0; sq < G::UNROLL; ++sq) {
    const int slot = (sq - nky + P + G::RING_DY) % G::RING_DY;            // ring slot of dY position t - ky + P, t = sq (mod 6)
    badr[sq] = keep_in_vgpr(lds_addr(dyring + slot * DROW + no * G::OST + lj * G::KST));
  }
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int units = a.B * units_per_img;
  __syncthreads();

  for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
    const int b = unit / units_per_img;
    const int q_lo = (unit - b * units_per_img) * band;
    const int rows = min(band, H - q_lo);            // band and q_lo are even
    const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<ST*>((const ST*)a.in + (long)b * a.in_bstride), 0, H * rowbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy.pool + (long)b * a.dy.pool_bstride), 0, Hp * Wp * nout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy.dpool + (long)b * a.dy.dpool_bstride), 0, Hp * Wp * nout * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t*>(a.dy.amax + (long)b * Hp * Wp * nout), 0, Hp * Wp * nout, 0x00020000);
    // position d of the band's stream <-> image row q_lo - P + d.  Before the first step (t = P): dY positions
    // 0 .. 2P and input positions P, P+1 are in LDS, input position P+2 and the cells of dY position 2P+1 in registers.
    const int y0 = q_lo - P;                          // image row of position 0 (even: P == 2)
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
      if (0 < rows) in_load(in_rs, q_lo);
      dy_issue(rp, rd, rc, y0 >> 1, 0);
      dy_issue(rp, rd, rc, (y0 >> 1) + 1, 1);
      if (0 < rows) in_store(P % G::RING_IN);
      if (1 < rows) in_load(in_rs, q_lo + 1);
      dy_conv(0, in_band(y0));
      dy_store(0, 0); dy_store(1, 1);
      dy_issue(rp, rd, rc, (y0 >> 1) + 2, 0);
      dy_conv(1, in_band(y0 + 2));
      dy_store(2, 0); dy_store(3, 1);
      if (1 < rows) in_store((P + 1) % G::RING_IN);
      if (2 < rows) in_load(in_rs, q_lo + 2);
      dy_conv(0, in_band(y0 + 4));
      dy_store(4, 0);                                 // position 5 (same cells) is stored by the first step
    }
    __syncthreads();

    for (int t0 = 0; t0 < rows + P; t0 += G::UNROLL) {
#pragma unroll
      for (int sq = 0; sq < G::UNROLL; ++sq) {
        const int t = t0 + sq;
        if (t < P) continue;                          // uniform
        if (t >= rows + P) break;
        // stage ahead: dY position t + P + 1 (its slot held position t - P - 1), input position t + 2
        {
          const int d = t + P + 1, y = y0 + d;
#ifndef DWKYO_ABL_NODY
          if (DENSE) {
            dense_store((sq + P + 1) % G::RING_DY);                       // requested one step ago
            dense_load(rdense, y + 1);
          } else {
            if ((d & 1) == 0) dy_conv(0, in_band(y));                     // requested one step ago
            dy_store((sq + P + 1) % G::RING_DY, (sq + P + 1) & 1);
            if ((d & 1) == 1) dy_issue(rp, rd, rc, (y + 1) >> 1, 0);      // next pooled row, used from the next step on
          }
#endif
#ifndef DWKYO_ABL_NOIN
          if (t + 2 - P < rows) in_store((sq + 2) % G::RING_IN);
          if (t + 3 - P < rows) in_load(in_rs, y0 + t + 3);
#endif
        }
        // multiply input position t with dY positions t - P .. t + P
        const int islot = sq % G::RING_IN;
        // operands of k-step st + LOOK are requested before the MFMAs of step st are issued (hipcc otherwise recycles
        // one register pair per load and waits for each load right before its use)
        constexpr int LOOK = 2;
        float av[LOOK + 1][MT];
        f32x4 bq[2];
        auto load_a = [&](int st, int set) {
          const int off = islot * ROWF * 4 + (4 * st) * CIN * 4;
          if (A64) {
#pragma unroll
            for (int mt = 0; mt < MT; mt += 2) {
              const f32x2 u = lds_load<f32x2>(aadr, off + 4 * mt);
              av[set][mt] = u.x; av[set][mt + 1] = u.y;
            }
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[set][mt] = lds_load<float>(aadr, off + 4 * mt);
          }
        };
        bq[0] = lds_load<f32x4>(badr[sq], 0);
#pragma unroll
        for (int st = 0; st < LOOK; ++st) load_a(st, st);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          if (st + LOOK < NS) load_a(st + LOOK, (st + LOOK) % (LOOK + 1));
          if ((st & 3) == 0 && st + 4 < NS) bq[((st >> 2) + 1) & 1] = lds_load<f32x4>(badr[sq], (st + 4) * 4);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt] = MFMA16(av[st % (LOOK + 1)][mt], bq[(st >> 2) & 1][st & 3], acc[mt]);
          __builtin_amdgcn_sched_barrier(0);
        }
#ifndef DWKYO_ABL_NOBAR
        __syncthreads();
#endif
      }
    }
  }

  // ---- one partial per workgroup: D tile mt holds rows m = MT*i + mt (i = 4 lj + r), column n
  float* part = a.partial + (long)blockIdx.x * a.pstride;
  const int nw = KS * G::KROW * nout;
#ifdef DWKYO_ABL_NOWRITE
  if (nvalid && no < nout && acc[0][0] == 123.456f) {
#else
  if (nvalid && no < nout) {
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = MT * (4 * lj + r) + mt;
        if (m < G::KROW) part[(nky * G::KROW + m) * nout + no] = acc[mt][r];
      }
    }
  }
  // bias gradient: per-thread cell sums -> LDS -> one thread per channel adds them in fixed order
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCELL; ++c) dbs[c * CONV_THREADS + tid] = cact[c] ? dbsum[c] : 0.f;
  __syncthreads();
  if (tid < nout) {
    float s = 0.f;
    for (int idx = tid; idx < Wp * nout; idx += nout) s += dbs[(idx / CONV_THREADS) * CONV_THREADS + (idx % CONV_THREADS)];
    part[nw + tid] = s;
  }
}

static inline int conv_dw_kyo_grid(cpp_ctx* ctx) { return ctx->num_cus * 4; }   // resident workgroups (4 per CU)

// rows per unit: split images into bands until the units fill the chip (4 workgroups per CU)
static inline int dw_kyo_band(int capacity, int B, int H) {
  int band = (H + 1) & ~1;
  while (B * ((H + band - 1) / band) < capacity && band > 8 && (band / 2) % 2 == 0) band /= 2;
  return band;
}

template <int CIN, int KS, int NS_, int IN_MODE, int CHB = 16, bool DENSE = false>
static inline int conv_dw_kyo_launch_t(cpp_ctx* ctx, const ConvArgsN& batch, int* grid_out) {
  typedef DwKyoGeom<CIN, KS, NS_> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)G::LDS_FLOATS * sizeof(float);
  auto kern = conv_dw_kyo_kernel<CIN, KS, NS_, IN_MODE, CHB, DENSE>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int capacity = conv_dw_kyo_grid(ctx) / batch.n;
  const int band = dw_kyo_band(capacity, a.B, a.H);
  const int upi = (a.H + band - 1) / band;
  const int units = a.B * upi;
  const int grid = units < capacity ? units : capacity;
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch, upi, band);
  LAUNCH_CHECK();
  *grid_out = grid;
  return 0;
}

int conv_dw_kyo_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, bool dense, const ConvArgsN& a, int* grid, bool* handled);
int conv_dw_kyo_dispatch_dense(cpp_ctx* ctx, int cin, int ks, int in_mode, int chb, const ConvArgsN& a, int* grid, bool* handled);

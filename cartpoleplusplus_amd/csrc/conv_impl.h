// Implicit-GEMM convolution kernels, (ky,(kx,c)) x o formulation -- the kernels of the first implementation, now the
// path for dW of the 3x3 layer and the fallback for shapes the (ky,o)-column kernels (conv_kyo.h, conv_dw_kyo.h) do
// not take (rows that cannot be staged as aligned 16-byte chunks, dX rows narrower than the strip grid).
// v_mfma_f32_16x16x4_f32 for the cartpole++ conv trunk
// (base_network.py:103-127: three slim.conv2d with 10 filters, stride 1, SAME, ReLU, each followed by
// a 2x2/2 VALID max-pool) and its backward.  Exact f32 (the MFMA is a k-ordered fmaf chain).
//
// Mapping (forward / dX):   D[pixel, o] += A[pixel, k] * B[k, o]
//   M tile  = 16 consecutive x positions of one output row          (MFMA rows,  lane & 15)
//   N tile  = the 10 output channels, padded to 16                  (MFMA cols,  lane & 15)
//   K       = (ky, (kx, c)); per ky the KS*CIN contiguous floats of an NHWC LDS row, padded to x4
// The weights live in VGPRs for the whole kernel (one B fragment per k-step, preloaded once per
// persistent workgroup); the A fragment is a single conflict-free ds_read_b32 from the LDS-staged,
// already whitened input tile (zero padding is applied in whitened space, base_network.py:97-99).
// A workgroup is 4 waves; wave w owns output rows 2w, 2w+1 of an 8-row tile, so the 2x2 max-pool,
// bias and ReLU happen in registers in the epilogue (the MFMA C layout gives every lane 4 consecutive
// x positions of one channel).
//
// Mapping (dW):             D[k', o] += A[k', pixel] * B[pixel, o]     (reduction over pixels, 4/MFMA)
//   one accumulator tile per (ky, 16 consecutive (kx,c)); B = dY rebuilt on the fly from the pooled
//   gradient + argmax code; accumulators persist across all tiles of a persistent workgroup and are
//   reduced deterministically (fixed order) through LDS and a second-stage kernel.
#pragma once
#include "common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int CONV_TH = 8;          // output rows per workgroup tile (4 waves x 2 rows) for the wide layers
// conv2 / conv3 (10 input channels, <= 32 columns) use 16-row tiles: each wave then owns two row pairs, which
// halves the per-tile staging / barrier overhead of these small layers (their tiles hold only ~260 MFMAs per wave)
constexpr int conv_th(int cin, int xtw) { return (cin == 10 && xtw <= 2) ? 16 : 8; }
constexpr int CONV_THREADS = 256;
constexpr int CONV_LDS_PAD = 16;    // floats; covers the k-padding over-read of the last tile row

// waves per SIMD the register allocator must leave room for: 2 workgroups per CU while the tile fits
// twice into the 160 KiB LDS, otherwise 1 (and up to 512 VGPRs)
constexpr int conv_wps(int cin, int ks, int xtw) {
  return ((conv_th(cin, xtw) + ks - 1) * (16 * xtw + ks - 1) * cin * 4 > 76 * 1024) ? 1 : 2;
}

// persistent workgroups per descriptor: the chip's resident capacity split over the batched networks, kept a
// multiple of 8 so every XCD gets a contiguous run of tiles
static inline int conv_grid_x(int capacity, int n, int ntiles) {
  int g = capacity / (n > 0 ? n : 1);
  if (g >= 8) g &= ~7;
  if (g < 1) g = 1;
  return g > ntiles ? ntiles : g;
}

// XCD-aware persistent tile order: workgroup b runs on XCD b % 8 (observed dispatch order, speed
// only); give every XCD a contiguous run of tiles so neighbouring row tiles of one image (which share
// 4 halo rows) hit the same 4 MiB L2.
__device__ __forceinline__ void conv_tile_range(int ntiles, int& start, int& step, int& end, int nb, int bid) {
  if ((nb & 7) == 0) {
    const int chunk = (ntiles + 7) >> 3;
    start = (bid & 7) * chunk + (bid >> 3);
    step = nb >> 3;
    end = min(((bid & 7) + 1) * chunk, ntiles);
  } else {
    start = bid; step = nb; end = ntiles;
  }
}

__device__ __forceinline__ float dy_value(const DyDesc& d, int b, int y, int x, int ch, int nch) {
  const int py = y >> 1, px = x >> 1;
  if (py >= d.Hp || px >= d.Wp) return 0.f;
  const long e = (long)(py * d.Wp + px) * nch + ch;
  const float g = d.dpool[(long)b * d.dpool_bstride + e];
  const int code = d.amax[(long)b * d.Hp * d.Wp * nch + e];
  return code == (4 | ((y & 1) * 2 + (x & 1))) ? g : 0.f;      // (bit 2: the pooled output was > 0 -- POOL_ACTIVE, conv_kyo.h)
}

template <int CIN, int KS, int XTW, int IN_MODE>
__device__ __forceinline__ void conv_stage_tile(float* lds, const ConvArgs& a, int b, int y0, int x0,
                                                int tid) {
  constexpr int P = KS / 2, TR = conv_th(CIN, XTW) + KS - 1, TC = 16 * XTW + KS - 1;
  constexpr int TILE = TR * TC * CIN;
  if (IN_MODE == IN_DY) {
    // dY tile rebuilt from POOLED cells: one (dpool, pool, amax) triple serves the 4 full-resolution positions
    // of its 2x2 window (4x fewer loads than per element).  The cells tile the plane, so every tile position is
    // written exactly once (zeros outside the image / where the arg-max is elsewhere).
    const DyDesc& d = a.dy;
    const int ty0 = y0 - P, tx0 = x0 - P;
    const int pr0 = ty0 >> 1, pc0 = tx0 >> 1;                       // floor, also for negatives
    constexpr int NPR = TR / 2 + 1, NPC = TC / 2 + 1;
    for (int idx = tid; idx < NPR * NPC * CIN; idx += CONV_THREADS) {
      const int c = idx % CIN;
      const int cell = idx / CIN;
      const int py = pr0 + cell / NPC, px = pc0 + cell % NPC;
      float gmv = 0.f; int code = -1;
      if (py >= 0 && py < d.Hp && px >= 0 && px < d.Wp) {
        const long e = (long)(py * d.Wp + px) * CIN + c;
        const int raw = d.amax[(long)b * d.Hp * d.Wp * CIN + e];
        gmv = (raw & 4) ? d.dpool[(long)b * d.dpool_bstride + e] : 0.f;      // (bit 2: POOL_ACTIVE, conv_kyo.h)
        code = raw & 3;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = 2 * py + (k >> 1) - ty0, col = 2 * px + (k & 1) - tx0;
        if (row >= 0 && row < TR && col >= 0 && col < TC)
          lds[(row * TC + col) * CIN + c] = (code == k) ? gmv : 0.f;
      }
    }
    return;
  }
  for (int idx = tid; idx < TILE; idx += CONV_THREADS) {
    const int c = idx % CIN;
    const int pc = idx / CIN;
    const int col = pc % TC;
    const int row = pc / TC;
    const int gy = y0 - P + row, gx = x0 - P + col;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
      if (IN_MODE == IN_F16_WHITEN) {
        const __half* src = (const __half*)a.in + (long)b * a.in_bstride;
        v = __half2float(src[((long)gy * a.W + gx) * CIN + c]) * a.scale[(long)b * a.white_bstride + c] + a.shift[(long)b * a.white_bstride + c];
      } else if (IN_MODE == IN_F32_WHITEN) {
        const float* src = (const float*)a.in + (long)b * a.in_bstride;
        v = src[((long)gy * a.W + gx) * CIN + c] * a.scale[(long)b * a.white_bstride + c] + a.shift[(long)b * a.white_bstride + c];
      } else {
        const float* src = (const float*)a.in + (long)b * a.in_bstride;
        v = src[((long)gy * a.W + gx) * CIN + c];
      }
    }
    lds[idx] = v;
  }
}


// ---------------------------------------------------------------------------------------------
// Vectorised, register-prefetched tile staging (f16 / f32 NHWC images or activations).
// A tile row [x0-P, x0-P+TC) x CIN is contiguous in memory AND in the LDS image, so a row is staged
// as aligned 16-byte chunks (8 halfs / 4 floats) with a per-element range test; the chunks of the NEXT
// tile are loaded into registers before the MFMA phase of the current one and only converted /
// whitened / written to LDS afterwards (issue-early / write-late), so HBM latency hides under compute.
// Positions outside the image are zero-filled by a separate pass (zero padding lives in whitened
// space, base_network.py:97-99).
// ---------------------------------------------------------------------------------------------
template <typename T> struct ChunkOps;
template <> struct ChunkOps<__half> {
  static constexpr int EPC = 8;
  static __device__ __forceinline__ float get_dyn(const uint4& v, int k) {      // k lane-dependent
    const uint32_t w = k < 4 ? (k < 2 ? v.x : v.y) : (k < 6 ? v.z : v.w);
    return __half2float(__ushort_as_half((unsigned short)((w >> ((k & 1) * 16)) & 0xffffu)));
  }
  static __device__ __forceinline__ float get(const uint4& v, int k) {
    const uint32_t w = k < 2 ? v.x : (k < 4 ? v.y : (k < 6 ? v.z : v.w));
    return __half2float(__ushort_as_half((unsigned short)((k & 1) ? (w >> 16) : (w & 0xffffu))));
  }
};
template <> struct ChunkOps<float> {
  static constexpr int EPC = 4;
  static __device__ __forceinline__ float get_dyn(const uint4& v, int k) {
    return __uint_as_float(k < 2 ? (k == 0 ? v.x : v.y) : (k == 2 ? v.z : v.w));
  }
  static __device__ __forceinline__ float get(const uint4& v, int k) {
    return __uint_as_float(k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)));
  }
};

template <int CIN, int KS, int XTW, typename T, bool WHITEN>
struct RowStager {
  static constexpr int P = KS / 2, TR = conv_th(CIN, XTW) + KS - 1, TC = 16 * XTW + KS - 1;
  static constexpr int EPC = ChunkOps<T>::EPC;
  static constexpr int CH = (TC * CIN + EPC - 1) / EPC + 1;    // chunks per tile row incl. alignment slack
  static constexpr int NCH = TR * CH;
  static constexpr int NV = (NCH + CONV_THREADS - 1) / CONV_THREADS;
  uint4 v[NV];

  // geometry of chunk `ch` of the tile at (y0, x0): returns false when the chunk holds nothing
  static __device__ __forceinline__ bool geom(const ConvArgs& a, int ch, int y0, int x0, int& r,
                                              int& a0, int& rowstart, int& rowend, int& gxs) {
    r = ch / CH;
    const int j = ch - r * CH;
    const int gy = y0 - P + r;
    if (ch >= NCH || gy < 0 || gy >= a.H) return false;
    gxs = max(x0 - P, 0);
    const int gxe = min(x0 - P + TC, a.W);
    rowstart = (gy * a.W + gxs) * CIN;
    rowend = (gy * a.W + gxe) * CIN;
    a0 = (rowstart & ~(EPC - 1)) + j * EPC;
    return a0 < rowend;
  }

  __device__ __forceinline__ void load(const ConvArgs& a, int b, int y0, int x0, int tid) {
    const T* img = (const T*)a.in + (long)b * a.in_bstride;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int r, a0, rs, re, gxs;
      v[i] = make_uint4(0u, 0u, 0u, 0u);
      if (geom(a, tid + CONV_THREADS * i, y0, x0, r, a0, rs, re, gxs))
        v[i] = *reinterpret_cast<const uint4*>(img + a0);
    }
  }

  // Lanes hold consecutive chunks (LDS addresses EPC dwords apart): a chunk that lies wholly inside its
  // row and lands 16-byte aligned in LDS goes out as ds_write_b128 (the common case: all but the two
  // chunks at the row ends); anything else falls back to per-element ds_write_b32.
  __device__ __forceinline__ void store(float* lds, const float2* wl, const ConvArgs& a, int y0, int x0,
                                        int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int r, a0, rs, re, gxs;
      if (geom(a, tid + CONV_THREADS * i, y0, x0, r, a0, rs, re, gxs)) {
        const int lbase = r * TC * CIN + (gxs - (x0 - P)) * CIN - rs;    // lds index = lbase + e
        int c = a0 % CIN;
        float x[EPC];
#pragma unroll
        for (int k = 0; k < EPC; ++k) {
          x[k] = ChunkOps<T>::get(v[i], k);
          if (WHITEN) { const float2 w = wl[c]; x[k] = x[k] * w.x + w.y; }
          c = (c + 1 == CIN) ? 0 : c + 1;
        }
        float* dst = lds + lbase + a0;
        if (a0 >= rs && a0 + EPC <= re && (((lbase + a0) & 3) == 0)) {
#pragma unroll
          for (int k = 0; k < EPC; k += 4)
            *reinterpret_cast<float4*>(dst + k) = make_float4(x[k], x[k + 1], x[k + 2], x[k + 3]);
        } else {
#pragma unroll
          for (int k = 0; k < EPC; ++k)
            if (a0 + k >= rs && a0 + k < re) dst[k] = x[k];
        }
      }
    }
  }

  // zero every tile position that lies outside the image
  static __device__ __forceinline__ void zero_halo(float* lds, const ConvArgs& a, int y0, int x0, int tid) {
    for (int p = tid; p < TR * TC; p += CONV_THREADS) {
      const int r = p / TC, col = p - r * TC;
      const int gy = y0 - P + r, gx = x0 - P + col;
      if (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) {
#pragma unroll
        for (int c = 0; c < CIN; ++c) lds[p * CIN + c] = 0.f;
      }
    }
  }
};

template <int IN_MODE> struct StageType { typedef float type; };
template <> struct StageType<IN_F16_WHITEN> { typedef __half type; };

// ---------------------------------------------------------------------------------------------
// forward / dX kernel
// ---------------------------------------------------------------------------------------------
template <int CIN, int KS, int XTW, int IN_MODE, int EPI>
__global__ __launch_bounds__(CONV_THREADS, conv_wps(CIN, KS, XTW)) void conv_fwd_kernel(const ConvArgsN batch) {
  const ConvArgs& a = batch.a[blockIdx.y];
  constexpr int TR = conv_th(CIN, XTW) + KS - 1, TCOLS = 16 * XTW, TC = TCOLS + KS - 1;
  constexpr int KROW = (KS * CIN + 3) / 4;
  constexpr int TILE = TR * TC * CIN;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;

  if (tid < CONV_LDS_PAD) lds[TILE + tid] = 0.f;

  // B fragments: wf[ky][kk] = W[ky][k' = 4kk + lj][o = li]  (0 outside the real K x N range)
  float wf[KS][KROW];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
    for (int kk = 0; kk < KROW; ++kk) {
      const int kp = 4 * kk + lj;
      float v = 0.f;
      if (kp < KS * CIN && li < a.nout) {
        if (IN_MODE == IN_DY || a.flip) {
          // dX = correlation of dY with the flipped, transposed weights:
          //   W'[ky][kx][o][c] = W[KS-1-ky][KS-1-kx][c][o],  W stored (KS,KS,Cin=nout,Cout=CIN)
          const int kx = kp / CIN, o = kp % CIN;
          v = a.w[(((KS - 1 - ky) * KS + (KS - 1 - kx)) * a.nout + li) * CIN + o];
        } else {
          v = a.w[(ky * KS * CIN + kp) * a.nout + li];
        }
        if (a.wscale != 0.f) v *= a.wscale;
      }
      wf[ky][kk] = v;
    }
  }
  float bias = 0.f;
  if (EPI == EPI_RELU_POOL && li < a.nout) bias = a.bias[li];

  const int Hp = a.H >> 1, Wp = a.W >> 1;
  const int tiles_per_img = a.tiles_x * a.tiles_y;
  int t_start, t_step, t_end;
  conv_tile_range(a.ntiles, t_start, t_step, t_end, gridDim.x, blockIdx.x);

  typedef typename StageType<IN_MODE>::type ST;
  constexpr bool WHITEN = (IN_MODE == IN_F16_WHITEN || IN_MODE == IN_F32_WHITEN);
  constexpr bool VEC = (IN_MODE != IN_DY);
  // prefetch the next tile's chunks into registers across the MFMA phase when the budget allows
  constexpr bool PREFETCH = VEC && (RowStager<CIN, KS, XTW, ST, WHITEN>::NV <= 8);
  RowStager<CIN, KS, XTW, ST, WHITEN> stg;
  float2* wl = reinterpret_cast<float2*>(lds + TILE + CONV_LDS_PAD);
  if (WHITEN && tid < CIN) wl[tid] = make_float2(a.scale[tid], a.shift[tid]);
  const bool vec = VEC && a.vec_ok;
  if (WHITEN) __syncthreads();
  if (PREFETCH && vec && t_start < t_end) {
    const int b = t_start / tiles_per_img;
    const int rem = t_start - b * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    stg.load(a, b, ty * conv_th(CIN, XTW), tx * TCOLS, tid);
  }

  for (int tile = t_start; tile < t_end; tile += t_step) {
    const int b = tile / tiles_per_img;
    const int rem = tile - b * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * conv_th(CIN, XTW), x0 = tx * TCOLS;

#ifdef CONV_ABLATE_NOSTAGE
    if (tile == t_start)
#endif
    {
    if (WHITEN && a.white_bstride != 0) {             // per-image statistics: this tile's image has its own table
      __syncthreads();
      if (tid < CIN) wl[tid] = make_float2(a.scale[(long)b * a.white_bstride + tid], a.shift[(long)b * a.white_bstride + tid]);
      __syncthreads();
    }
    if (vec) {
      if (!PREFETCH) stg.load(a, b, y0, x0, tid);
      stg.store(lds, wl, a, y0, x0, tid);
      RowStager<CIN, KS, XTW, ST, WHITEN>::zero_halo(lds, a, y0, x0, tid);
    } else {
      conv_stage_tile<CIN, KS, XTW, IN_MODE>(lds, a, b, y0, x0, tid);
    }
    }
    __syncthreads();
#ifndef CONV_ABLATE_NOSTAGE
    if (PREFETCH && vec && tile + t_step < t_end) {
      const int nt = tile + t_step;
      const int nb = nt / tiles_per_img;
      const int nrem = nt - nb * tiles_per_img;
      const int nty = nrem / a.tiles_x, ntx = nrem - nty * a.tiles_x;
      stg.load(a, nb, nty * conv_th(CIN, XTW), ntx * TCOLS, tid);
    }
#endif

#pragma unroll 1
    for (int rp = 0; rp < conv_th(CIN, XTW) / 8; ++rp) {     // row pairs of this wave (2 for the 16-row tiles)
    const int wrow = wave + 4 * rp;
    const int yrow = y0 + 2 * wrow;
    if (yrow < a.H) {   // wave-uniform
      f32x4 acc[2][XTW];
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < XTW; ++t) acc[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

      const float* lp = lds + (2 * wrow) * TC * CIN + li * CIN + lj;
      // tile row q = r + ky feeds output row 0 with W[ky=q] and output row 1 with W[ky=q-1]
#pragma unroll
#ifdef CONV_ABLATE_NOMFMA
      for (int q = 0; q < 1; ++q) {
#else
      for (int q = 0; q < KS + 1; ++q) {
#endif
#pragma unroll
        for (int kk = 0; kk < KROW; ++kk) {
#pragma unroll
          for (int t = 0; t < XTW; ++t) {
            const float av = lp[(q * TC + t * 16) * CIN + 4 * kk];
            if (q < KS) acc[0][t] = MFMA16(av, wf[q < KS ? q : 0][kk], acc[0][t]);
            if (q >= 1) acc[1][t] = MFMA16(av, wf[q >= 1 ? q - 1 : 0][kk], acc[1][t]);
          }
        }
      }

      if (EPI == EPI_RELU_POOL) {
        const int py = yrow >> 1;
        if (py < Hp && li < a.nout) {
#pragma unroll
          for (int t = 0; t < XTW; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int px = ((x0 + t * 16 + 4 * lj) >> 1) + h;
              const float z00 = acc[0][t][2 * h] + bias, z01 = acc[0][t][2 * h + 1] + bias;
              const float z10 = acc[1][t][2 * h] + bias, z11 = acc[1][t][2 * h + 1] + bias;
              float m = z00; int code = 0;
              if (z01 > m) { m = z01; code = 1; }
              if (z10 > m) { m = z10; code = 2; }
              if (z11 > m) { m = z11; code = 3; }
              if (px < Wp) {
                const long e = (long)(py * Wp + px) * a.nout + li;
                a.out[(long)b * a.out_bstride + e] = fmaxf(m, 0.f);
                a.out_amax[(long)b * Hp * Wp * a.nout + e] = (uint8_t)(code | (m > 0.f ? 4 : 0));      // (bit 2: POOL_ACTIVE, conv_kyo.h)
              }
            }
          }
        }
      } else {
        if (li < a.nout) {
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int y = yrow + r;
            if (y < a.H) {
#pragma unroll
              for (int t = 0; t < XTW; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int x = x0 + t * 16 + 4 * lj + i;
                  if (x < a.W)
                    a.out[(long)b * a.out_bstride + ((long)y * a.W + x) * a.nout + li] = acc[r][t][i];
                }
              }
            }
          }
        }
      }
    }
    }   // row pairs
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// dW / db kernel
// ---------------------------------------------------------------------------------------------
template <int CIN, int KS>
struct DwGeom {
  static constexpr int KT = (KS * CIN + 15) / 16;   // 16-row accumulator tiles per ky
  static constexpr int NT = KS * KT;
};

// dY of a tile is kept in LDS at POOLED resolution: gm = (pool > 0 ? dpool : 0) and the argmax code,
// (CONV_TH/2) x (TCOLS/2) cells x DW_GP dwords; the two output rows of a wave share one pooled row, so
// one gm + one code read give the B operands of both rows.
#ifndef DW_PREFETCH
#define DW_PREFETCH 1
#endif
#ifndef DW_STEP_UNROLL
#define DW_STEP_UNROLL 1
#endif
constexpr int DW_GP = 20;     // dwords per pooled cell: 16 channels + pad so lane groups 8 px apart miss each other's banks

// sched_group_barrier wants literal constants: unroll the per-tile-row pattern through templates
template <int Q, int KS, int KT>
struct DwSched {
  static __device__ __forceinline__ void emit() {
    if constexpr (Q + 2 < KS + 1) __builtin_amdgcn_sched_group_barrier(0x100, (KT + 1) / 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, (Q == 0 || Q == KS) ? KT : 2 * KT, 0);
    if constexpr (Q < KS) DwSched<Q + 1, KS, KT>::emit();
  }
};

template <int XTW, int TH>
struct DyStager {
  static constexpr int PR = TH / 2, PC = 8 * XTW;
  static constexpr int NE_MAX = PR * PC * CPP_NOUT_MAX;
  static constexpr int NV = (PR * PC * 10 + CONV_THREADS - 1) / CONV_THREADS;   // nout <= 10 fast path
  float g[NV], pv[NV];
  int code[NV];

  __device__ __forceinline__ void load(const ConvArgs& a, int b, int y0, int x0, int tid) {
    const DyDesc& d = a.dy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + CONV_THREADS * i;
      const int o = idx % a.nout, pp = idx / a.nout;
      const int pc = pp % PC, pr = pp / PC;
      const int py = (y0 >> 1) + pr, px = (x0 >> 1) + pc;
      g[i] = 0.f; pv[i] = 0.f; code[i] = 255;
      if (pr < PR && py < d.Hp && px < d.Wp) {
        const long e = (long)(py * d.Wp + px) * a.nout + o;
        g[i] = d.dpool[(long)b * d.dpool_bstride + e];
        const int raw = d.amax[(long)b * d.Hp * d.Wp * a.nout + e];
        pv[i] = (raw & 4) ? 1.f : 0.f;      // (bit 2: POOL_ACTIVE, conv_kyo.h)
        code[i] = raw & 3;
      }
    }
  }
  __device__ __forceinline__ void store(float* gm, unsigned char* gc, const ConvArgs& a, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + CONV_THREADS * i;
      const int o = idx % a.nout, pp = idx / a.nout;
      if (pp < PR * PC) {
        gm[pp * DW_GP + o] = pv[i] > 0.f ? g[i] : 0.f;
        gc[pp * DW_GP + o] = (unsigned char)code[i];
      }
    }
  }
};

// (bx, by, gx): the workgroup's place in a (gx, networks) grid -- blockIdx / gridDim for a launch of its own, a slice of the
// grid when the kernel shares a launch with another one (conv3_bwd_pair.hip)
template <int CIN, int KS, int XTW, int IN_MODE>
__device__ __forceinline__ void conv_dw_body(const ConvArgsN& batch, const int bx, const int by, const int gx) {
  const ConvArgs& a = batch.a[by];
  constexpr int TR = conv_th(CIN, XTW) + KS - 1, TCOLS = 16 * XTW, TC = TCOLS + KS - 1;
  constexpr int TILE = TR * TC * CIN;
  constexpr int KT = DwGeom<CIN, KS>::KT, NT = DwGeom<CIN, KS>::NT;
  constexpr int SP = TCOLS >= 32 ? 8 : 4;          // pixel spacing inside one MFMA step (bank spread)
  constexpr int GMF = (conv_th(CIN, XTW) / 2) * (TCOLS / 2) * DW_GP;   // floats of the gm image
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;

  if (tid < CONV_LDS_PAD) lds[TILE + tid] = 0.f;
  float2* wl = reinterpret_cast<float2*>(lds + TILE + CONV_LDS_PAD);
  float* gm = lds + TILE + CONV_LDS_PAD + 2 * CIN;
  unsigned char* gc = reinterpret_cast<unsigned char*>(gm + GMF);
  // zero the channel padding of gm / gc once (lanes li >= nout read it)
  for (int i = tid; i < GMF; i += CONV_THREADS) { gm[i] = 0.f; gc[i] = 255; }

  f32x4 acc[KS][KT];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int t = 0; t < KT; ++t) acc[ky][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;

  const int tiles_per_img = a.tiles_x * a.tiles_y;
  int t_start, t_step, t_end;
  conv_tile_range(a.ntiles, t_start, t_step, t_end, gx, bx);

  typedef typename StageType<IN_MODE>::type ST;
  constexpr bool WHITEN = (IN_MODE == IN_F16_WHITEN || IN_MODE == IN_F32_WHITEN);
  constexpr bool PREFETCH = (RowStager<CIN, KS, XTW, ST, WHITEN>::NV <= 8) && (DW_PREFETCH != 0);
  RowStager<CIN, KS, XTW, ST, WHITEN> stg;
  DyStager<XTW, conv_th(CIN, XTW)> dst;
  if (WHITEN && tid < CIN) wl[tid] = make_float2(a.scale[tid], a.shift[tid]);
  const bool vec = a.vec_ok;
  const bool dyfast = a.nout <= 10;
  __syncthreads();
  if (t_start < t_end) {
    const int b = t_start / tiles_per_img;
    const int rem = t_start - b * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    if (PREFETCH && vec) stg.load(a, b, ty * conv_th(CIN, XTW), tx * TCOLS, tid);
  }

  for (int tile = t_start; tile < t_end; tile += t_step) {
    const int b = tile / tiles_per_img;
    const int rem = tile - b * tiles_per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * conv_th(CIN, XTW), x0 = tx * TCOLS;

    if (dyfast) dst.load(a, b, y0, x0, tid);      // issued first: their latency hides under the tile stores
    if (WHITEN && a.white_bstride != 0) {             // per-image statistics: this tile's image has its own table
      __syncthreads();
      if (tid < CIN) wl[tid] = make_float2(a.scale[(long)b * a.white_bstride + tid], a.shift[(long)b * a.white_bstride + tid]);
      __syncthreads();
    }
    if (vec) {
      if (!PREFETCH) stg.load(a, b, y0, x0, tid);
      stg.store(lds, wl, a, y0, x0, tid);
      RowStager<CIN, KS, XTW, ST, WHITEN>::zero_halo(lds, a, y0, x0, tid);
    } else {
      conv_stage_tile<CIN, KS, XTW, IN_MODE>(lds, a, b, y0, x0, tid);
    }
    if (dyfast) {
      dst.store(gm, gc, a, tid);
    } else {       // generic (nout > 10): straight from global
      for (int idx = tid; idx < (conv_th(CIN, XTW) / 2) * (TCOLS / 2) * a.nout; idx += CONV_THREADS) {
        const int o = idx % a.nout, pp = idx / a.nout;
        const int pc = pp % (TCOLS / 2), pr = pp / (TCOLS / 2);
        const int py = (y0 >> 1) + pr, px = (x0 >> 1) + pc;
        float gv = 0.f; int cd = 255;
        if (py < a.dy.Hp && px < a.dy.Wp) {
          const long e = (long)(py * a.dy.Wp + px) * a.nout + o;
          const int raw = a.dy.amax[(long)b * a.dy.Hp * a.dy.Wp * a.nout + e];
          gv = (raw & 4) ? a.dy.dpool[(long)b * a.dy.dpool_bstride + e] : 0.f;      // (bit 2: POOL_ACTIVE, conv_kyo.h)
          cd = raw & 3;
        }
        gm[pp * DW_GP + o] = gv; gc[pp * DW_GP + o] = (unsigned char)cd;
      }
    }
    __syncthreads();
    if (tile + t_step < t_end) {
      const int nt = tile + t_step;
      const int nb = nt / tiles_per_img;
      const int nrem = nt - nb * tiles_per_img;
      const int nty = nrem / a.tiles_x, ntx = nrem - nty * a.tiles_x;
      if (PREFETCH && vec) stg.load(a, nb, nty * conv_th(CIN, XTW), ntx * TCOLS, tid);
    }

#pragma unroll 1
    for (int rp = 0; rp < conv_th(CIN, XTW) / 8; ++rp) {     // row pairs of this wave
    const int wrow = wave + 4 * rp;
    const int yrow = y0 + 2 * wrow;
    if (yrow < a.H) {
#pragma unroll DW_STEP_UNROLL
      for (int st = 0; st < TCOLS / 4; ++st) {
        const int xbase = (st / SP) * 4 * SP + (st % SP);        // smallest of the step's 4 pixels
        if (x0 + xbase >= a.W) continue;                          // wave-uniform
        const int xloc = xbase + SP * lj;
        const int cell = (wrow * (TCOLS / 2) + (xloc >> 1)) * DW_GP + li;
        const float gv = gm[cell];
        const int cd = gc[cell];
        const float b0 = (cd == (xloc & 1)) ? gv : 0.f;           // row 2w   : code = 0*2 + (x&1)
        const float b1 = (cd == 2 + (xloc & 1)) ? gv : 0.f;       // row 2w+1 : code = 1*2 + (x&1)
        bsum += b0 + b1;
        const float* lp = lds + ((2 * wrow) * TC + xloc) * CIN + li;
        // issue the A reads of two tile rows ahead of the MFMAs that consume them (the scheduler otherwise
        // recycles one register pair and exposes the LDS latency after every 4 MFMAs)
        float av[KS + 1][KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) { av[0][t] = lp[16 * t]; av[1][t] = lp[TC * CIN + 16 * t]; }
#pragma unroll
        for (int q = 0; q < KS + 1; ++q) {
          if (q + 2 < KS + 1) {
#pragma unroll
            for (int t = 0; t < KT; ++t) av[q + 2][t] = lp[(q + 2) * TC * CIN + 16 * t];
          }
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            if (q < KS) acc[q < KS ? q : 0][t] = MFMA16(av[q][t], b0, acc[q < KS ? q : 0][t]);
            if (q >= 1) acc[q >= 1 ? q - 1 : 0][t] = MFMA16(av[q][t], b1, acc[q >= 1 ? q - 1 : 0][t]);
          }
        }
        // pin the software pipeline in the emitted code: [reads of rows 0,1 + the dY cell] then per tile row
        // [reads of row q+2][MFMAs of row q]
        __builtin_amdgcn_sched_group_barrier(0x100, KT + 2, 0);
        DwSched<0, KS, KT>::emit();
      }
    }
    }   // row pairs
    __syncthreads();
  }

  // cross-wave reduction in fixed order (wave 0, 1, 2, 3) through LDS, then one partial per block
  float* red = lds;   // NT*256 floats (launch sizes the LDS for max(tile image, NT*256 + 64))
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int idx = ((ky * KT + t) * 64 + lane) * 4 + i;
            if (w == 0) red[idx] = acc[ky][t][i]; else red[idx] += acc[ky][t][i];
          }
    }
    __syncthreads();
  }
  // db: sum the 4 pixel lanes of each channel, then the waves
  bsum += __shfl_xor(bsum, 16);
  bsum += __shfl_xor(bsum, 32);
  float* bred = lds + NT * 256;
  if (lj == 0) bred[wave * 16 + li] = bsum;
  __syncthreads();

  float* part = a.partial + (long)bx * a.pstride;
  const int nw = KS * KS * CIN * a.nout;
  for (int e = tid; e < nw; e += CONV_THREADS) {
    const int o = e % a.nout;
    const int kfull = e / a.nout;            // ky*KS*CIN + k'
    const int ky = kfull / (KS * CIN), kp = kfull - ky * (KS * CIN);
    const int t = kp >> 4, row = kp & 15;
    const int ln = (row >> 2) * 16 + o;      // D[row][col]: lane = (row/4)*16 + col, reg = row%4
    part[e] = red[((ky * KT + t) * 64 + ln) * 4 + (row & 3)];
  }
  if (tid < a.nout) part[nw + tid] = (bred[tid] + bred[16 + tid]) + (bred[32 + tid] + bred[48 + tid]);
}

// second stage: grad[e] = sum_blocks partial[blk][e], fixed order (conv_dw_l23.hip)
struct DwReduceBatch { DwReduceDesc d[DW_REDUCE_MAX]; int block_start[DW_REDUCE_MAX + 1]; int n; };

// second stage of the conv dW kernels: 64 outputs x NS slices of the partial list per workgroup (64 NS threads); slices are
// combined in fixed order.  One launch serves every queued (layer, network) reduction: workgroup -> (descriptor, output
// block) via a prefix table.  `red`: [NS][64] floats of LDS.
template <int NS, int TS = NS>      // NS slices of the partial list, walked by TS thread slices (64 TS threads): same sums for any TS
__device__ __forceinline__ void conv_dw_reduce_body(const DwReduceBatch& rb, const int bx, float (*red)[64]) {
  static_assert(NS % TS == 0, "whole slices per thread slice");
  int p = 0;
  while (p + 1 < rb.n && bx >= rb.block_start[p + 1]) ++p;
  const DwReduceDesc d = rb.d[p];
  const int el = threadIdx.x & 63, tslice = threadIdx.x >> 6;
  const int e = (bx - rb.block_start[p]) * 64 + el;
  const int n = d.nw + d.nout;
  const int per = (d.nblocks + NS - 1) / NS;
#pragma unroll
  for (int q = 0; q < NS / TS; ++q) {
    const int slice = tslice * (NS / TS) + q;
    const int b0 = slice * per, b1 = min(b0 + per, d.nblocks);
    float s = 0.f;
    if (e < n) {
      int b = b0;
      // (32 partials in flight, then 8, then one by one -- added in list order whatever the batch: the same bits; a slice of 32 partials
      // -- conv1's 512 workgroups over 16 slices -- is one round trip instead of four)
      for (; b + 32 <= b1; b += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = d.partial[(long)(b + u) * d.pstride + e];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
      }
      for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = d.partial[(long)(b + u) * d.pstride + e];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; b < b1; ++b) s += d.partial[(long)b * d.pstride + e];
    }
    red[slice][el] = s;
  }
  __syncthreads();
  if (tslice == 0) {
    float t = 0.f;
    if (e < n) {
#pragma unroll
      for (int k = 0; k < NS; ++k) t += red[k][el];     // fixed order
      if (e < d.nw) d.grad_w[e] = t; else d.grad_b[e - d.nw] = t;
    }
    if (d.sq_part) {                                 // (uniform) the block's share of the gradient list's squared norm
      double sq = (double)t * (double)t;
      for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
      if (el == 0) d.sq_part[bx - rb.block_start[p]] = sq;
    }
  }
}
int launch_dw_reduce_batch(cpp_ctx* ctx, const DwReduceBatch& rb, const StatsRide* st = nullptr);

template <int CIN, int KS, int XTW, int IN_MODE, int EPI>
static inline int conv_fwd_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  const ConvArgs& a = batch.a[0];
  constexpr int TR = conv_th(CIN, XTW) + KS - 1, TC = 16 * XTW + KS - 1;
  const size_t lds_bytes = (size_t)(TR * TC * CIN + CONV_LDS_PAD + 2 * CIN) * sizeof(float);
  auto kern = conv_fwd_kernel<CIN, KS, XTW, IN_MODE, EPI>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  const int grid = conv_grid_x(ctx->num_cus * per_cu, batch.n, a.ntiles);
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

template <int CIN, int KS>
static inline int conv_dw_grid(cpp_ctx* ctx, int xtw_max) {
  (void)xtw_max;
  return ctx->num_cus * 2;
}

template <int CIN, int KS, int XTW, int IN_MODE>
__global__ __launch_bounds__(CONV_THREADS, conv_wps(CIN, KS, XTW))
__attribute__((amdgpu_waves_per_eu(conv_wps(CIN, KS, XTW), conv_wps(CIN, KS, XTW)))) void conv_dw_kernel(const ConvArgsN batch) {
  conv_dw_body<CIN, KS, XTW, IN_MODE>(batch, blockIdx.x, blockIdx.y, gridDim.x);
}

template <int CIN, int KS, int XTW, int IN_MODE>
static inline int conv_dw_launch_t(cpp_ctx* ctx, const ConvArgsN& batch, int* grid_out) {
  const ConvArgs& a = batch.a[0];
  constexpr int TR = conv_th(CIN, XTW) + KS - 1, TC = 16 * XTW + KS - 1;
  constexpr int NT = DwGeom<CIN, KS>::NT;
  constexpr int GMF = (conv_th(CIN, XTW) / 2) * (8 * XTW) * DW_GP;
  size_t fl = (size_t)(TR * TC * CIN + CONV_LDS_PAD + 2 * CIN) + GMF + GMF / 4 + 4;
  if (fl < (size_t)NT * 256 + 64) fl = (size_t)NT * 256 + 64;
  const size_t lds_bytes = fl * sizeof(float);
  auto kern = conv_dw_kernel<CIN, KS, XTW, IN_MODE>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  const int grid = conv_grid_x(ctx->num_cus * per_cu, batch.n, a.ntiles);
  *grid_out = grid;
  if (ctx->pair && ctx->pair->layer == 2 && CIN == 10 && KS == 3 && XTW == 1 && IN_MODE == IN_F32_PLAIN) {      // leaves with conv3's dX (conv3_bwd_pair.hip)
    ctx->pair->dw = batch; ctx->pair->dw_gx = grid; ctx->pair->dw_lds = lds_bytes; ctx->pair->have_dw = true;
    return 0;
  }
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

// per-translation-unit dispatchers (each .hip file instantiates a slice of the template space so the
// big fully-unrolled kernels compile in parallel)
int conv_fwd_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, int epi, const ConvArgsN& a);
int conv_fwd_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, int epi, const ConvArgsN& a);
int conv_dw_dispatch_l1(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a, int* grid);
int conv_dw_dispatch_l23(cpp_ctx* ctx, int cin, int ks, int xtw, int in_mode, const ConvArgsN& a, int* grid);

// conv1 forward on the f16 matrix pipes with f32-grade operands ("k16"): v_mfma_f32_16x16x32_f16.
//
// The images of this path are 8-bit renders held as f16 (replay_memory.py:32, bullet_cartpole.py:239-243), so the A
// operand of conv1 is EXACT in f16 as long as it is the raw pixel.  The whitening of base_network.py:95-99 is affine
// per channel, x_w = s_c (x - mu_c) inside the image (mu_c = -t_c / s_c of the f32 table) and 0 in the SAME padding, so it
// moves into the B operand -- CHUNK BY CHUNK (round 4).  An MFMA contracts 32 k values; a chunk holds RK = 30 real k = (kx, c)
// of an input row and two synthetic "ones" slots whose A value is a constant 2^T and whose weights are minus the chunk's own
// sum_k V'_k mu_c(k):
//
//   z[y,x,o] = b_o + sum_{ky} sum_{chunks} [ sum_{k in chunk} V'_k x~_k  -  sum_{k in chunk} V'_k mu_c(k) ],   V' = the pieces of W s
//
// Every MFMA therefore adds a sum of WHITENED terms V' (x - mu) to the f32 accumulator -- the partial sums stay of the size
// of the result, as in a convolution of the whitened tensor (which is what the reference's TF kernels accumulate), instead of
// growing to sum |V mu| and cancelling once per row: on a render's near-constant channels (a sky: s = 9, a blind camera
// with one glint: s = 990) that intermediate reaches 10^1 ... 10^3 times the result and its roundings stay in it (rounds 1-3
// kept ONE ones channel per row behind k = 90: 3.8e-4 on conv1 outputs where float32 numpy is 1.0e-4 from float64, r04_render_probe.txt).
// 3 x 30 = 90 = KS * CIN for the 5x5x18 layer, no extra MFMA (two chunks less than the 32 + ones layout for 6 and 12 channels);
// chunk starts stay 4-byte aligned (60 bytes).  The ones weights are computed in f64 FROM THE ROUNDED PIECES V' (so the two
// halves of a chunk cancel to the rounding of V', not of W s) and carry four pieces: slot 30 the first two (three) of
// -sum V' mu 2^(S-T), slot 31 the next -- 44 bits of a number that can be 10^3 times a weight.
// The SAME padding: a tap left / right of the image must contribute V' * 0, but its ones share is already in the chunk; the A
// element of such a tap is therefore the PIVOT p_c = f16(mu_c) instead of 0 (same mask instruction: and + or), which leaves
// V'_k (p_c - mu_c), at most 2^-12 of a chunk term -- and that remainder is data independent: E[row class][x class][o], a
// 5 x 4 x NO table the epilogue subtracts from the four border columns.  Rows outside the image are not walked at all.
// The f32 weights V = W s are split into F16_PIECES f16 pieces
// V 2^S = h + m (+ l) (h = f16(V 2^S), m = f16(rest), l = f16(rest); S puts the largest |V| just under 2^15), every
// f16 x f16 product is exact in the f32 accumulator, and the MFMAs of a k chunk add  x * (h + m (+ l)).  With three pieces
// (CPP_PRECISION_EXACT) that is the f32-accumulated sum of exact products of the SAME operands the f32 kernel uses -- no
// operand rounded to fewer than its 24 bits (pieces below the f16 normal range keep an absolute 2^-24-S floor, < 1e-9 of the
// largest weight); with two (CPP_PRECISION_FAST, the default) every weight is within 2^-22 of itself, about one f32 ulp (F16_PIECES below).  Two / three
// 16-cycle MFMAs replace eight 32-cycle ones per 32 k values.
//
// Everything else is the (ky,o)-column formulation of conv_kyo.h: one input row per step, KS in-flight output rows
// in the accumulators, rotating weights read from LDS at per-lane addresses, bias folded into the accumulator reset
// (scaled by 2^S; the epilogue multiplies the pooled value by 2^-S), pool pairs through a wave-private LDS buffer.
//
// The A operand does not go through LDS: a lane's 8 consecutive k values are 16 contiguous bytes of the image row
// (k = (kx, c) runs along the row), but only 4-byte aligned (36 bytes per pixel), and a misaligned ds_read_b128 is
// slow (measured: the 6 A reads of a row cost more than its 36 B reads).  buffer_load_dwordx4 takes 4-byte aligned
// addresses at full rate, so every lane loads its operands of the NEXT row straight from global memory (L1/L2 hits:
// the windows of neighbouring lanes overlap) right after the MFMAs that consumed the current ones.  No row staging,
// no barrier in the row loop: the waves of a workgroup only share the weight image.  Elements outside the image
// (left / right SAME padding, the over-read past k = KROW) are cleared with per-lane AND masks, which only the
// waves' border tiles apply; reads may touch up to 128 bytes before and 256 after an image (the arena pads).
#pragma once
#include <type_traits>
#include "conv_kyo.h"
#include "conv3_img.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 k16_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned k16_u32x4 __attribute__((ext_vector_type(4)));

// ---- B16 mode (conv2: f32 activations in).  An f32 is exactly the sum of three bf16 numbers (8 + 8 + 8 significand bits, f32's
// exponent range: no scaling); the previous layer's epilogue writes its pooled output as three bf16 planes next to the
// f32 tensor, this kernel loads the planes like conv1 loads pixels, the weights are split the same way and all 3 x 3 exact
// products are issued (nine 16-cycle MFMAs instead of eight 32-cycle ones per 32 k values; no ones channel, no whitening).
// Which of the nine piece products are issued.  With round-to-nearest pieces x = h + m + l has |m| <= 2^-8 |x| and |l| <= 2^-17 |x|
// (a piece's remainder is at most half its last place), so the products m*l, l*m and l*l together are at most 2^-24 |x y| -- half an
// f32 ulp of the product, worst case; each output then takes K = 250 f32 accumulation steps that round by as much of the running sum.
// Against the float64 oracle the six-product result is as close as the nine-product one (pooled conv2 output 4.47e-6 / 4.47e-6 abs
// at magnitude 7.5, weight gradient 5.8e-7 / 6.0e-7 rel; the f32-input MFMA kernels: 4.77e-6, 7.5e-7;
// profiles/experiments/r03_b16_products.txt).  The kernels' B16 / ORDER template value is the largest i + j (h = 0, m = 1, l = 2)
// still issued: B16_SIX under CPP_PRECISION_FAST, B16_NINE under CPP_PRECISION_EXACT (and with CPP_B16_PRODUCTS=9 in the ablation build).
//
// F16_PIECES: f16 pieces of conv1's f32 operand (weights in the forward kernel, dY in conv_dw16.h; the other operand is the raw f16
// pixel, exact).  Two round-to-nearest pieces h + m leave |x - h - m| <= 2^-22 |x| (11 + 11 significand bits; the sign of m is one more):
// the operand to within about ONE f32 ulp; a third piece holds what is left, at most two bits.  Against the float64 oracle at 64x64x18, B = 256: pooled conv1
// output 3.2e-6 (two) / 3.7e-6 (three) / 7.0e-6 (f32-input MFMA) max abs at magnitude 7.1, conv1 weight gradient 1.22e-6 / 1.19e-6 /
// 1.69e-6 rel (profiles/experiments/r03_f16_pieces.txt).  CPP_PRECISION_FAST: 2; CPP_PRECISION_EXACT: 3 (every product exact).
//
// Both arithmetic contracts are in the release library; cpp_ctx_set_precision (include/cartpolepp_abi.h) chooses per context:
// CPP_PRECISION_FAST (default): two f16 pieces, six bf16 products; CPP_PRECISION_EXACT: three / nine -- every product exact.
#define F16_PIECES 2
#define F16_PIECES_EXACT 3
#define B16_SIX 2
#define B16_NINE 4
static inline int b16_order(const cpp_ctx* ctx) {
  static const int sw = cpp_switch_int("CPP_B16_PRODUCTS", 6) == 9 ? B16_NINE : B16_SIX;      // (ablation build only)
  return (ctx && ctx->precision == CPP_PRECISION_EXACT) ? B16_NINE : sw;
}
static inline bool f16_exact(const cpp_ctx* ctx) { return ctx && ctx->precision == CPP_PRECISION_EXACT; }
__device__ __forceinline__ unsigned k16_bf16_bits(float x) {        // round-to-nearest-even bf16 of a finite f32
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void k16_split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {   // x = h + m + l exactly
  const unsigned hb = k16_bf16_bits(x);
  const float r1 = x - __uint_as_float(hb << 16);
  const unsigned mb = k16_bf16_bits(r1);
  const float r2 = r1 - __uint_as_float(mb << 16);
  h = (unsigned short)hb; m = (unsigned short)mb; l = (unsigned short)(__float_as_uint(r2) >> 16);     // r2 has <= 8 significant bits
}

// ---- A-operand loads the compiler does not see (K16_ASYNC_A).  The compiler's own s_waitcnt placement for this loop is
// vmcnt(0) in front of every chunk's first operand use: the pooled-row writer's stores are issued under per-lane / uniform
// conditions, so the pass cannot count them, and a wave sat out the full round trip of ten stores every second row.  With the
// loads issued from inline asm the pass has nothing to wait for; the waits are placed by hand as vmcnt(N), N = the number of
// vector-memory instructions issued AFTER the loads that are needed (they retire in order) -- which is a compile-time number
// because every such instruction is issued unconditionally (disabled / out-of-image stores are dropped by their buffer
// descriptor's range check instead of being branched around).
// THE COUNTS HOLD FOR THE INSTRUCTION ORDER OF LLVM'S DEFAULT SCHEDULER.  Built with -mllvm -amdgpu-sched-strategy=iterative-ilp this
// translation unit gives results that differ from run to run (4 of 10 runs of tests/test_gpu_distributed.py's cfg3 comparison;
// profiles/experiments/r04_sched_strategy.txt): do not change the scheduling strategy of conv_fwd_k16.hip without re-deriving them.
typedef int k16_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ k16_i32x4 k16_raw_desc(const void* base, int num_records) {      // raw buffer, stride 0 (as make_buffer_rsrc)
  const unsigned long long b = (unsigned long long)base;
  return (k16_i32x4){(int)(unsigned)b, (int)((b >> 32) & 0xFFFFu), num_records, 0x00020000};
}
__device__ __forceinline__ void k16_issue_b128(k16_u32x4& dst, const k16_i32x4& desc, int voff, int soff) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
__device__ __forceinline__ void k16_issue_b32(unsigned& dst, const k16_i32x4& desc, int voff, int soff) {
  asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(desc), "s"(soff) : "memory");
}
// s_waitcnt vmcnt(N) that the uses of `x...` cannot be moved in front of
template <int N> __device__ __forceinline__ void k16_wait_vm(k16_u32x4& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void k16_wait_vm(unsigned& x) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(x) : "n"(N) : "memory"); }

template <int CIN, int KS, int XT, int IPW, bool B16 = false, int NPCS = F16_PIECES>      // (geometry: B16 mode or not; f16 pieces of a weight)
struct K16Geom {
  static constexpr int NO = KYO_NO;
  static constexpr int P = KS / 2;
  static constexpr int NT = (KS * NO + 15) / 16;
  static constexpr int STRIPS = 4 / IPW, SW = 16 * XT, WPAD = STRIPS * SW;
  static constexpr int KROW = KS * CIN;                    // real k = (kx, c) of one input row
  static constexpr int RK = B16 ? 32 : 30;                 // real k values per MFMA chunk; f16 mode: slots 30, 31 of every chunk are the ones slots
  static constexpr int NPA = B16 ? 3 : 1;                  // planes of the A operand
  static constexpr int NCH = (KROW + RK - 1) / RK;         // MFMA k chunks per row
  static constexpr int NPC = B16 ? 3 : NPCS;               // f16 / bf16 pieces of a weight
  // weight image in LDS: slab (chunk, piece) holds the 16-byte operand (ky, lane group g, o) at ky*PS + g*GS + o*16;
  // the padded strides keep the rotating per-lane reads of a ds_read_b128 at 1.2 accesses per bank quad (2.45 compact)
  static constexpr bool PADDED = NCH * NPC * 5632 <= 64 * 1024;
  static constexpr int PS = PADDED ? 1120 : 4 * NO * 16, GS = PADDED ? 256 : NO * 16, SLAB = PADDED ? 5632 : KS * 4 * NO * 16;
  static constexpr int WLB = NCH * NPC * SLAB;             // bytes
  static constexpr int EF = 2 * 2 * 8 * XT * NO * 2;       // floats per wave: (value, code) of the two rows of a pool pair, two pairs (the writer of pair r runs under the rows of pair r + 1)
  // conv2's instance for 32x32 inputs can run conv3 as its tail: the two pooled 16x16 images of the workgroup, zero-haloed
  static constexpr int N3 = (B16 && XT == 1 && IPW == 2) ? C3_IPW * C3_IMGF * 4 + 16 : 0;
  // f16 mode: the border table E [2 P + 1 row classes][NO][x = 0, 1, W - 2, W - 1] (floats, in accumulator units) and the pivots [CIN] (halves)
  static constexpr int NRC = 2 * P + 1;
  static constexpr int CT_BYTES = B16 ? 0 : NRC * NO * 16 + ((CIN * 2 + 15) & ~15);
  static constexpr int LDS_BYTES = WLB + 4 * EF * 4 + 64 + N3 + CT_BYTES;
  static constexpr int BIAS_BYTES = 128;                   // the buffer descriptor starts this far before the image
  // element e of lane group g in chunk ch is slot kl = 8 g + e of the chunk: real k = RK ch + kl if kl < RK (and k < KROW)
  // can dword v of chunk ch in M tile m (of any strip, any lane) ever hold an element that must be cleared or replaced?  The
  // synthetic slots (ones, zero fill), a tap left of the image (kx < P: only a strip's first tile can touch x < 0)
  // or a tap right of it (kx > P: any tile, the image may end anywhere in a strip).
  static __host__ __device__ constexpr bool vgpr_may_need_mask(int ch, int m, int v) {
    for (int g = 0; g < 4; ++g)
      for (int h = 0; h < 2; ++h) {
        const int kl = 8 * g + 2 * v + h;
        if (kl >= RK) return true;
        const int k = RK * ch + kl;
        if (k >= KROW) return true;
        const int kx = k / CIN;
        if (kx > P) return true;
        if (kx < P && m == 0) return true;
      }
    return false;
  }
};

#ifndef K16_WGS
#define K16_WGS 2
#endif
#ifndef K16_ROTATE_PRIO
#define K16_ROTATE_PRIO 1
#endif
template <int CIN, int KS, int XT, int IPW, bool PLAIN = false, int B16 = 0, int NPCS = F16_PIECES>      // B16: 0, or B16_SIX / B16_NINE
__global__ __launch_bounds__(CONV_THREADS, K16_WGS) void conv_fwd_k16_kernel(const ConvArgsN batch) {
  typedef K16Geom<CIN, KS, XT, IPW, (B16 != 0), NPCS> G;
  static_assert(B16 != 0 || KS == 5, "the border table of the f16 mode is written for 5x5 (P = 2)");
  constexpr int P = G::P, NT = G::NT, NCH = G::NCH, NPC = G::NPC, NPA = G::NPA, NO = KYO_NO;
  constexpr bool ODD = (CIN & 1) != 0;            // 2-byte aligned operand windows: 20 bytes from the aligned address below + a funnel shift
#if defined(K16_CLOCK_PROBE)
  const unsigned long long pe0 = __builtin_amdgcn_s_memrealtime();
#endif
  const ConvArgs& a = batch.a[blockIdx.y];
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* wl = lds_raw;                                    // weight image
  float2* ebuf = reinterpret_cast<float2*>(lds_raw + G::WLB);     // [4 waves][2 pairs][2 parities][8*XT][NO]
  float* red = reinterpret_cast<float*>(lds_raw + G::WLB + 4 * G::EF * 4);
  float* img3 = reinterpret_cast<float*>(lds_raw + G::WLB + 4 * G::EF * 4 + 64);      // (G::N3 > 0 only)
  const bool fuse3 = G::N3 > 0 && a.n3_w != nullptr;                                  // uniform; the launcher checked the geometry
  if (fuse3)
    for (int i = threadIdx.x; i < C3_IPW * C3_IMGF / 4; i += CONV_THREADS) reinterpret_cast<float4*>(img3)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int strip = wave % G::STRIPS;
  // bands of output rows (launcher: a.nbands = 2 when a launch would leave half of the workgroup slots empty): band k of an image
  // computes output rows [k band_rows, (k + 1) band_rows); its row walk starts P input rows earlier -- band_rows - P is a multiple of KS,
  // so the unrolled loop's static ring positions hold -- and the output rows in front of the band complete as partial sums and are dropped
  const int nbands = a.nbands > 1 ? a.nbands : 1;
  const int band = (int)blockIdx.x % nbands;
  const int b0 = ((int)blockIdx.x / nbands) * IPW;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;
  const int ymin = band * a.band_rows;                                  // first output row of the band (0 without bands)
  const int qbeg = band > 0 ? ymin - P : 0;                             // first input row walked (a multiple of KS)
  const int qend = (nbands > 1 && band + 1 < nbands ? ymin + a.band_rows : H) + P;      // one past the last step

  // (the accumulators' bias is requested with the weights: after the weight image it was one more L2 round trip, ~1 us, in-kernel clock)
  float braw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = 16 * t + li;
    braw[t] = (!PLAIN && j < KS * NO && j % NO < nout) ? a.bias[j % NO] : 0.f;
  }
  // ---- one-time setup: the split weight image
  float sc, inv;                                      // 2^S, 2^-S
  int onesT = 0;                                      // f16 mode: the ones slots' A value is 2^T
  float* ctab = reinterpret_cast<float*>(lds_raw + G::WLB + 4 * G::EF * 4 + 64 + G::N3);      // [NRC][NO][4] (f16 mode)
  unsigned short* pivot = reinterpret_cast<unsigned short*>(ctab + G::NRC * NO * 4);            // [CIN] f16 bits
  {
    // A thread owns UNITS: the 8 consecutive slots kl = 8 g + e of one (ky, chunk, lane group g, o) -- the 16 bytes one lane's
    // ds_read_b128 fetches.  It loads their weights, splits them, stores each piece with ONE 16-byte LDS write, and (f16 mode) keeps
    // the pieces' exact sum V' in registers for its share of the chunk's ones sum and of the border sums: nothing is read back.
    // (Round 4's first version wrote 2-byte pieces and re-read them in a serial f64 loop per sum: 12.9 us of setup per workgroup,
    // 2.7 us of it the split and 4.5 us the sums -- -DK16_SETUP_PROBE.)
    constexpr int NU = KS * NCH * 4 * NO;             // units, o fastest, then g, chunk, ky
    constexpr int NUW = (NU + CONV_THREADS - 1) / CONV_THREADS;
    float wv[NUW][8];
    float vmax = 0.f;
    // scratch in the pool-pair buffers (not in use yet): mu_c [CIN], the units' ones partials [NU] (f64) and border partials [NU][2]
    constexpr int NB = 7 / CIN + 2;                   // taps kx a unit's 8 consecutive k values can touch (2 from 8 channels up)
    double* mu = reinterpret_cast<double*>(ebuf);
    double* opart = mu + ((CIN + 1) & ~1);
    float* bpart = reinterpret_cast<float*>(opart + (B16 ? 0 : NU));      // [NU][NB]
    float* scl = bpart + (B16 ? 0 : NU * NB);         // [CIN] whitening scale, [CIN] p_c - mu_c
    float* dmu = scl + CIN;
    static_assert(B16 || (((CIN + 1) & ~1) + NU) * 8 + (NU * NB + 2 * CIN) * 4 <= 4 * G::EF * 4, "the setup's scratch fits the pool-pair buffers");
    // branch-free: every load of the build is in flight before the first use
#pragma unroll
    for (int n = 0; n < NUW; ++n) {
      const int u = tid + n * CONV_THREADS;
      const int o = u % NO, g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH, ky = u / (4 * NO * NCH);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kl = 8 * g + e, k = G::RK * ch + kl;
        const bool real = u < NU && o < nout && kl < G::RK && k < G::KROW;
        const float w = a.w[real ? (ky * G::KROW + k) * nout + o : 0];
        wv[n][e] = real ? w : 0.f;
      }
    }
    if (!B16) {
      if (tid < CIN) {                                // mu_c = -t_c / s_c: x s + t = s (x - mu); a constant channel (table (0, 0): stats_body.h) has V' = 0
        const float s_c = a.scale[tid], t_c = a.shift[tid];
        const double m_c = s_c != 0.f ? -(double)t_c / (double)s_c : 0.0;
        const _Float16 p_c = (_Float16)(float)m_c;
        mu[tid] = m_c;
        pivot[tid] = __builtin_bit_cast(unsigned short, p_c);
        scl[tid] = s_c;
        dmu[tid] = (float)((double)(float)p_c - m_c);
      }
      __syncthreads();                                // (the weight loads are still in flight: 24 scattered scale loads per thread less)
    }
#pragma unroll
    for (int n = 0; n < NUW; ++n) {
      const int u = tid + n * CONV_THREADS;
      const int g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = wv[n][e];
        if (!B16) v *= scl[(G::RK * ch + 8 * g + e) % CIN];      // (slots past the row: w = 0)
        if (a.wscale != 0.f) v *= a.wscale;
        wv[n][e] = v;
        vmax = fmaxf(vmax, fabsf(v));
      }
    }
#ifdef K16_CLOCK_PROBE
    const unsigned long long pq0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0) red[wave] = vmax;
    __syncthreads();
#ifdef K16_CLOCK_PROBE
    if (tid == 0 && (blockIdx.x % 97) == 5 && blockIdx.y == 1) printf("K16PRE loads+max %llu, to sync %llu\n", pq0 - pe0, __builtin_amdgcn_s_memrealtime() - pe0);
#endif
    vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int S = 0;
    if (!B16 && vmax > 0.f && vmax < 3.0e38f) S = 14 - ilogbf(vmax);       // vmax 2^S in [2^14, 2^15); bf16 pieces need no scale
    S = S > 100 ? 100 : (S < -100 ? -100 : S);
    sc = ldexpf(1.f, S); inv = ldexpf(1.f, -S);
#pragma unroll
    for (int n = 0; n < NUW; ++n) {
      const int u = tid + n * CONV_THREADS;
      const int o = u % NO, g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH, ky = u / (4 * NO * NCH);
      unsigned short pcs[3][8];
      double po = 0.0;                                // - sum_e V'_e mu_c(e)            (2^S units)
      float pb[NB];                                   // sum_e V'_e (p_c - mu_c) per tap kx the unit touches (first tap: kx0)
#pragma unroll
      for (int b = 0; b < NB; ++b) pb[b] = 0.f;
      const int kx0 = (G::RK * ch + 8 * g) / CIN;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = wv[n][e] * sc;
        if (B16) k16_split3(v, pcs[0][e], pcs[1][e], pcs[2][e]);
        else {
          const _Float16 hh = (_Float16)v;
          const float r1 = v - (float)hh;
          const _Float16 mm = (_Float16)r1;
          const float r2 = r1 - (float)mm;
          const _Float16 ll = (_Float16)r2;
          pcs[0][e] = __builtin_bit_cast(unsigned short, hh); pcs[1][e] = __builtin_bit_cast(unsigned short, mm); pcs[2][e] = __builtin_bit_cast(unsigned short, ll);
          const int kl = 8 * g + e, k = G::RK * ch + kl;
          if (kl < G::RK && k < G::KROW) {            // (uniform per e within a (ch, g) class; padded o / u: V' = 0)
            const double vp = NPC > 2 ? ((double)(float)hh + (double)(float)mm) + (double)(float)ll : (double)(float)hh + (double)(float)mm;      // exact
            const int c = k % CIN;
            po -= vp * mu[c];
            const float d = (float)vp * dmu[c];
            const int b = k / CIN - kx0;
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) pb[bb] += bb == b ? d : 0.f;
          }
        }
      }
      if (u < NU) {
        const uint32_t dst = lds_addr(wl + ky * G::PS + g * G::GS + o * 16);
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc)
          lds_store(dst, (ch * NPC + pc) * G::SLAB, (k16_u32x4){(unsigned)pcs[pc][0] | ((unsigned)pcs[pc][1] << 16), (unsigned)pcs[pc][2] | ((unsigned)pcs[pc][3] << 16),
                                                               (unsigned)pcs[pc][4] | ((unsigned)pcs[pc][5] << 16), (unsigned)pcs[pc][6] | ((unsigned)pcs[pc][7] << 16)});
        if (!B16) {
          opart[u] = po;
#pragma unroll
          for (int b = 0; b < NB; ++b) bpart[NB * u + b] = pb[b];
        }
      }
    }
    if (!B16) {
      // ---- the ones slots and the border table from the units' partials, in a fixed order (g = 0 .. 3; chunk, g ascending)
      __syncthreads();
      constexpr int NJ1 = KS * NCH * NO;
      constexpr int NJW = (NJ1 + CONV_THREADS - 1) / CONV_THREADS;
      double osumv[NJW];                              // this thread's ones sums (job = tid + k * 256), kept for the slot writes below
      double omax = 0.0;
#pragma unroll
      for (int jk = 0; jk < NJW; ++jk) {              // -sum_{k in chunk} V'_k mu_c(k)
        const int job = tid + jk * CONV_THREADS;
        osumv[jk] = 0.0;
        if (job < NJ1) {
          const int o = job % NO, cy = job / NO;      // cy = ky * NCH + ch
          const double* pp = opart + (cy * 4) * NO + o;
          const double acc = ((pp[0] + pp[NO]) + pp[2 * NO]) + pp[3 * NO];
          osumv[jk] = acc;
          omax = fmax(omax, fabs(acc));
        }
      }
      // border table E[rc][o][xi] (accumulator units): output row class rc (rows 0 .. P-1, interior, rows H-P .. H-1) sees the input
      // rows ky with 0 <= y + ky - P < H; x = 0: taps kx = 0, 1 are left of the image; x = 1: kx = 0; x = W - 2: kx = KS - 1;
      // x = W - 1: kx = KS - 2, KS - 1 (P = 2).  Each is the sum over those (ky, kx) of sum_c V'[ky,kx,c,o] (p_c - mu_c), i.e. of the
      // units' border partials that belong to tap kx.
      for (int job = tid; job < G::NRC * NO * 4; job += CONV_THREADS) {
        const int xi = job & 3, o = (job >> 2) % NO, rc = job / (4 * NO);
        const int kxlo = xi == 0 ? 0 : (xi == 1 ? 0 : (xi == 2 ? KS - 1 : KS - 2)), kxhi = xi == 0 ? 1 : (xi == 1 ? 0 : KS - 1);
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const bool seen = rc < P ? ky >= P - rc : (rc > P ? ky <= KS - 1 - (rc - P) : true);
          float kacc = 0.f;
          { float& acc = kacc;
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int kxa = (G::RK * ch + 8 * g) / CIN;      // (compile-time: the unit's first tap)
              const float* bp = bpart + NB * ((((ky * NCH + ch) * 4) + g) * NO + o);
#pragma unroll
              for (int b = 0; b < NB; ++b) acc += (kxa + b >= kxlo && kxa + b <= kxhi) ? bp[b] : 0.f;      // (selects, not branches: every load in flight)
            }
          }
          acc += seen ? kacc : 0.f;
        }
        ctab[job] = acc;
      }
      for (int o = 32; o > 0; o >>= 1) omax = fmax(omax, __shfl_xor(omax, o));
      if (lane == 0) red[4 + wave] = (float)omax;      // (an upper bound is all that is needed: rounded up below)
      __syncthreads();
      const float om = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])) * 1.0001f;
      onesT = (om > 0.f && om < 3.0e38f) ? ilogbf(om) - 14 : 0;      // |ones weight| 2^-T < 2^15
      onesT = onesT < 0 ? 0 : (onesT > 15 ? 15 : onesT);      // (T > 15: |mu| > 2^10 -- not an image; the pieces saturate to inf and the output says so)
#pragma unroll
      for (int jk = 0; jk < NJW; ++jk) {
        const int job = tid + jk * CONV_THREADS;
        if (job < NJ1) {
          const int o = job % NO, ch = (job / NO) % NCH, ky = job / (NO * NCH);
          double v = ldexp(osumv[jk], -onesT);
          unsigned char* dst = wl + ky * G::PS + 3 * G::GS + o * 16 + 6 * 2;      // slot 30 = (lane group 3, element 6), slot 31 behind it
#pragma unroll
          for (int slot = 0; slot < 2; ++slot)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) {
              const _Float16 hh = (_Float16)(float)v;
              v -= (double)(float)hh;
              *reinterpret_cast<unsigned short*>(dst + slot * 2 + (ch * NPC + pc) * G::SLAB) = __builtin_bit_cast(unsigned short, hh);
            }
        }
      }
    }
  }
  __syncthreads();                                   // weight image visible; no barrier after this one
#ifdef K16_CLOCK_PROBE
  const unsigned long long pe1 = __builtin_amdgcn_s_memrealtime();
#endif

  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int simg = swave / G::STRIPS, sstrip = swave % G::STRIPS;
  const int sbimg = b0 + simg;
  if (sbimg >= a.B) return;                          // wave-uniform

  // ---- per-lane column bookkeeping (as conv_kyo.h): column j = 16 t + li = p * NO + o; row loop unrolled by KS
  uint32_t wadr[KS][NT];
  uint32_t eadr[NT];
  float biast[NT];
  float2* ev = ebuf + swave * (2 * 2 * 8 * XT * NO);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = 16 * t + li;
    const bool valid = j < KS * NO;
    const int o = j % NO;
    biast[t] = braw[t] * sc;
    eadr[t] = PLAIN ? (uint32_t)(((sstrip * G::SW + 4 * lj) * nout + o) * 4)
                    : keep_in_vgpr(lds_addr(ev + (lj * 2) * NO + o));
    const int p = valid ? j / NO : 0;
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int ky = (sq + P - p + KS) % KS;
      wadr[sq][t] = keep_in_vgpr(lds_addr(wl + ky * G::PS + lj * G::GS + (valid ? o : 0) * 16));
    }
  }

  // ---- operand masks: dword v of a lane's 16-byte window of chunk ch in tile m becomes (u & emask) | ecst.  emask clears the
  // synthetic slots of the lane's group (ones, zero fill) and the taps of the SAME padding (pixel x + kx - P outside [0, W));
  // ecst puts the ones slots' constant 2^T there, and the channel's pivot p_c = f16(mu_c) into the padding taps (f16 mode: the
  // chunk's ones slots subtract V' mu for every k of the chunk, the pivot leaves V' (p_c - mu_c), the epilogue's table removes that).
  // Applied unconditionally wherever K16Geom::vgpr_may_need_mask says a mask can matter at all (a run-time "is this a
  // border tile" made the compiler compute both variants and select: 48 VALU per row).
  unsigned emask[NCH][XT][4], ecst[NCH][XT][4];
  const unsigned ones_bits = (unsigned)(15 + onesT) << 10;      // f16 2^T
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int m = 0; m < XT; ++m) {
        unsigned cs = 0u, bm = 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int kl = 8 * lj + 2 * v + h;
          const int k = G::RK * ch + kl;
          if (kl >= G::RK) { if (!B16) cs |= ones_bits << (16 * h); continue; }
          if (k >= G::KROW) continue;
          const int kx = k / CIN;
          const int xi = strip * G::SW + m * 16 + li + kx - P;
          const bool inside = xi >= 0 && xi < W;
          if (inside) bm |= 0xFFFFu << (16 * h);
          else if (!B16) cs |= (unsigned)pivot[k % CIN] << (16 * h);
        }
        ecst[ch][m][v] = cs;
        emask[ch][m][v] = bm;
      }
  // ---- f16 mode: the border columns' table (see the file comment).  Accumulator element r of tile m is pixel x = strip SW + 16 m + 4 lj + r;
  // pool pair h = r / 2.  Left: x = 0, 1 = (strip 0, m 0, lj 0, h 0); right: x = W - 2, W - 1 = (strip xr / SW, m (xr % SW) / 16, lj (xr % 16) / 4,
  // h (xr % 4) / 2), xr = W - 2 (W is even).  fl / fr[m][h]: 1.0 on the lanes that hold such a pair, else 0.0 -- the epilogue
  // subtracts f * E with an FMA instead of branching.
  uint32_t ctadr[NT];
  float fl = 0.f, fr[XT][2];
  {
    const int xr = W - 2;
    const bool lgrp = !B16 && sstrip == 0 && lj == 0;
    const bool rgrp = !B16 && sstrip == xr / G::SW && lj == (xr % 16) / 4;
    fl = lgrp ? 1.f : 0.f;
#pragma unroll
    for (int m = 0; m < XT; ++m)
#pragma unroll
      for (int h = 0; h < 2; ++h) fr[m][h] = (rgrp && m == (xr % G::SW) / 16 && h == (xr % 4) / 2) ? 1.f : 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) ctadr[t] = keep_in_vgpr(lds_addr(ctab + (((16 * t + li) % NO) << 2)));
  }

  // ---- pooled-row writer: a lane owns PAIRS (o, o+1) of the wave's 8*XT pooled columns (nout is even: dispatch) -- one
  // 8-byte value store, one 2-byte code store and three 4-byte bf16-plane stores per pair
  constexpr int NC = (8 * XT * (NO / 2) + 63) / 64;
  uint32_t cadr[NC], c3adr[NC];
  bool cact[NC];
  unsigned coe[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = lane + 64 * i;
    const int hn = nout >> 1;
    const int xl = idx / hn, o = 2 * (idx - xl * hn);
    const int px = ((sstrip * G::SW) >> 1) + xl;
    cact[i] = idx < 8 * XT * hn && px < Wp;
    cadr[i] = keep_in_vgpr(lds_addr(ev + (cact[i] ? xl * NO + o : 0)));
    coe[i] = cact[i] ? (unsigned)(px * nout + o) : 0x3FFFFFFCu;      // inactive lanes: beyond every descriptor's range (x 1, 2, 4)
    c3adr[i] = lds_addr(cact[i] ? img3 + simg * C3_IMGF + (C3_PW + px + 1) * C3_C + o : img3 + C3_IPW * C3_IMGF);   // (junk pair behind the images)
  }
  constexpr int SH = 5;                               // vector-memory stores of one writer HALF (all issued, every row; see k16_issue_b128)
  const __amdgpu_buffer_rsrc_t b16_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_b16 ? a.out_b16 + (long)sbimg * (Hp * Wp * nout) : (unsigned short*)a.out, 0,
      a.out_b16 ? (int)(2 * a.out_b16_plane * 2 + (long)Hp * Wp * nout * 2) : 0, 0x00020000);      // this image in the three planes, no further
  const int b16_plane_bytes = (int)(a.out_b16_plane * 2);
  // (a null output -- the target networks' f32 pool1 and codes in the fused step -- gets an empty range: its stores are issued
  // and dropped, so the number of vector-memory instructions per writer pass does not depend on the network)
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out ? a.out + (long)sbimg * a.out_bstride : (float*)a.out_b16, 0, (a.out ? (PLAIN ? H * W : Hp * Wp) * nout * 4 : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t amax_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_amax ? a.out_amax + (long)sbimg * Hp * Wp * nout : (uint8_t*)a.out_b16, 0, (a.out_amax ? Hp * Wp * nout : 0), 0x00020000);

  // ---- A operands: 16 bytes per lane and (tile, chunk) straight from the image row
  const int rowbytes = W * CIN * 2;
  void* const in_base = (void*)((const char*)(B16 ? (const void*)a.in_b16 : a.in) + ((long)(a.img_slot ? a.img_slot[sbimg] : sbimg) * a.in_bstride) * 2 - G::BIAS_BYTES);
  const int in_records = (B16 ? 2 * (int)a.plane_stride : 0) + H * rowbytes + G::BIAS_BYTES + 256;      // this image (B16: in its three planes) + the masked overhang
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(in_base, 0, in_records, 0x00020000);
  const k16_i32x4 in_desc = k16_raw_desc(in_base, in_records);      // the same descriptor for the inline-asm loads
  const int avoff0 = G::BIAS_BYTES + ((strip * G::SW + li - P) * CIN + 8 * lj) * 2;     // >= 128 - 2 P CIN
  const int avoff = ODD ? (avoff0 & ~3) : avoff0;
  const unsigned ashift = ODD ? (unsigned)(avoff0 & 2) : 0u;    // per-lane constant: 16 * m * CIN pixels further keeps the parity
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  k16_u32x4 av[NPA][NCH][XT];
  unsigned ax[NCH][XT];                               // ODD: the fifth dword of the window
  const int plane_bytes = B16 ? (int)a.plane_stride : 0;      // bytes between the bf16 planes of the input
#ifdef K16_NO_ASYNC_A
  constexpr bool ASYNC_A = false;
#else
  constexpr bool ASYNC_A = !PLAIN;                    // (the plain-output epilogue stores a data-dependent number of rows)
#endif
  // Round 6: up to three chunks per row the loads are builtins the compiler counts itself (as conv_rs16.h's; every vector-memory
  // instruction of the row loop is unconditional, so its own s_waitcnt placement is exact) and nothing below is hand-counted -- conv2
  // forward (B16 mode: 55.7 -> 54.7 us at cfg3, 302 -> 290 at cfg5) and conv1 forward of the geometries conv_rs16.h does not take
  // (50 x 50 renders, EXACT mode).  Five chunks (30 channels, cfg5's conv1) keep the inline-asm loads and the vmcnt(N) below: there
  // the compiler's waits come out as vmcnt(0 .. 3) where 13 loads and stores may stay in flight, and 6 registers spill
  // (1190 -> 1342 us; profiles/experiments/r06_k16_counted_ab.sh).  `make check-waits` still analyses the whole listing.
#ifdef K16_ASM_LOADS
  constexpr bool COUNTED_A = false;
#else
  constexpr bool COUNTED_A = NCH <= 3;
#endif
  constexpr int LPC = XT * NPA + (ODD ? XT : 0);      // vector-memory loads per chunk and row
  auto load_a = [&](int ch, int y) {
#pragma unroll
    for (int m = 0; m < XT; ++m) {
      if (ASYNC_A && !COUNTED_A) {
#pragma unroll
        for (int pa = 0; pa < NPA; ++pa)
          k16_issue_b128(av[pa][ch][m], in_desc, avoff, pa * plane_bytes + y * rowbytes + (m * 16 * CIN + G::RK * ch) * 2);
        if (ODD) k16_issue_b32(ax[ch][m], in_desc, avoff, y * rowbytes + (m * 16 * CIN + G::RK * ch) * 2 + 16);
        continue;
      }
#pragma unroll
      for (int pa = 0; pa < NPA; ++pa) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, avoff + (m * 16 * CIN + G::RK * ch) * 2, pa * plane_bytes + y * rowbytes, 0);
        av[pa][ch][m] = (k16_u32x4){v.x, v.y, v.z, v.w};
      }
      if (ODD) ax[ch][m] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, avoff + (m * 16 * CIN + G::RK * ch) * 2 + 16, y * rowbytes, 0);
    }
  };

  f32x4 acc[XT][NT];
#pragma unroll
  for (int m = 0; m < XT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) load_a(ch, qbeg);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0, 0x00020000);      // empty range: stores are dropped
  const bool wr_f32 = PLAIN || a.out != nullptr, wr_code = a.out_amax != nullptr;     // target networks of the fused step: bf16 planes only
  // ---- the pooled-row writer, one HALF per row (round 4).  Pair r of output rows is complete at the end of step 2 r + 1; its pooled
  // row is written under the MFMAs of the next two rows: lane set i = 0 (64 of the 8 XT NO / 2 channel pairs) right behind chunk 0's
  // MFMAs of step 2 r + 2, set i = 1 (the rest) behind chunk 0's of step 2 r + 3 -- in the shadow of 16 MFMAs the wave has just
  // issued, instead of as a phase between two rows in which the wave issues none (26 us of the 107 us launch; -DK16_ABL_NOEPI),
  // and every row issues exactly SH stores: no dummy stores to keep the hand-counted waits static (7.5 us; -DK16_ABL_NODUMMY).
  // The (value, code) pairs of a pair of rows therefore live for two more rows: two buffer sets, alternating.
  // half: 0 / 1 = the parity of the output row this step completes; pr: the pooled row (< 0: none -- the stores go to an
  // out-of-range offset and are dropped, the instruction count stays).
  f32x4 wtop = {0.f, 0.f, 0.f, 0.f}, wbot = {0.f, 0.f, 0.f, 0.f};      // the half's (value, code) x 2 of the pair's two rows, requested at the top of the step
  auto writer_load = [&](const int half, const int pr, const bool sure = false) {      // sure: 0 <= pr < Hp is known (interior rows)
    if (PLAIN || half >= NC) return;
    const int i = half < NC ? half : 0;
    const int prs = (sure || (pr >= 0 && pr < Hp)) ? pr : 0;
    const uint32_t ca = cadr[i] + (uint32_t)((prs & 1) * (2 * 8 * XT * NO) * 8);
    wtop = lds_load<f32x4>(ca, 0); wbot = lds_load<f32x4>(ca, (8 * XT * NO) * 8);
  };
  auto writer_half = [&](const int half, const int pr, const bool sure = false) {
    if (PLAIN) return;
    // (one lane set per pooled row -- NC = 1, 16-pixel strips: the other row of the pair has nothing to write, but issues the SAME five
    // stores, out of every descriptor's range.  Rounds 3-4 issued five stores to an empty descriptor on a path of their own; one store
    // group on every path is what lets cartpoleplusplus_amd/csrc/tools/check_async_loads.py count the hand-placed waits on the kernel's flow graph.)
    const bool dummy = half >= NC;
    const bool live = !dummy && (sure || (pr >= 0 && pr < Hp));           // uniform
    if (!ASYNC_A && !live) return;
    const int i = half < NC ? half : 0;
    if (ASYNC_A || cact[i]) {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const int prs = live ? pr : 0;
      const unsigned co = live ? coe[i] : 0x3FFFFFFCu;
      const int orow = prs * Wp * nout;
      float pv[2] = {0.f, 0.f}; int code[2] = {0, 0};
      if (live || !ASYNC_A) {
        const f32x4 top = wtop, bot = wbot;            // (writer_load)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool lower = bot[2 * e] > top[2 * e];
          const float mx = lower ? bot[2 * e] : top[2 * e];
          code[e] = (lower ? 2 + __float_as_int(bot[2 * e + 1]) : __float_as_int(top[2 * e + 1])) | (mx > 0.f ? POOL_ACTIVE : 0);
          pv[e] = mx > 0.f ? mx * inv : 0.f;
        }
      }
      if (fuse3 && live) lds_store(c3adr[i], prs * (C3_PW * C3_C * 4), (f32x2){pv[0], pv[1]});      // conv3's input row, in LDS
      if (ASYNC_A || wr_f32) __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(pv[0]), __float_as_uint(pv[1])}, out_rsrc, (int)(co * 4), orow * 4, 0);
      if (!B16 && (ASYNC_A || a.out_b16)) {    // the next layer's A operand: three bf16 planes of the same tensor (conv1's output only: B16 mode is
                                               // conv2's forward, whose output feeds conv3 as f32 -- round 6: its rows no longer split and store three dropped planes)
        // truncating split (cheaper than round-to-nearest in this MFMA-issue-bound loop, equally exact: the pieces are
        // the value's three consecutive byte-groups of significand, x = h + m + l)
        unsigned hb[2], mb[2], lb[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          hb[e] = __float_as_uint(pv[e]) & 0xFFFF0000u;
          const float r1 = pv[e] - __uint_as_float(hb[e]);
          mb[e] = __float_as_uint(r1) & 0xFFFF0000u;
          lb[e] = __float_as_uint(r1 - __uint_as_float(mb[e]));
        }
        __builtin_amdgcn_raw_buffer_store_b32((hb[0] >> 16) | hb[1], b16_rsrc, (int)(co * 2), orow * 2, 0);
        __builtin_amdgcn_raw_buffer_store_b32((mb[0] >> 16) | mb[1], b16_rsrc, (int)(co * 2), b16_plane_bytes + orow * 2, 0);
        __builtin_amdgcn_raw_buffer_store_b32((lb[0] >> 16) | (lb[1] & 0xFFFF0000u), b16_rsrc, (int)(co * 2), 2 * b16_plane_bytes + orow * 2, 0);
      }
      if (ASYNC_A || wr_code) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(code[0] | (code[1] << 8)), amax_rsrc, (int)co, orow, 0);
    }
  };

  // B operands of ONE k chunk.  The pieces are consumed small-to-large (pc = NPC-1 .. 0); as soon as the MFMAs of a piece are
  // issued its registers are reloaded with the same piece of the NEXT chunk (the next row's first chunk after the last one), so
  // every read has the other pieces' MFMAs (>= 16 x 16 cycles) to land: the latency cover of a second register set without
  // its 48 VGPRs (and without the register-to-register moves an odd chunk count needed).
  f16x8 bv[NT][NPC];
  auto load_b_piece = [&](int ch, int pc, const uint32_t* wa) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bv[t][pc] = lds_load<f16x8>(wa[t], (ch * NPC + pc) * G::SLAB);
    }
  };
#pragma unroll
  for (int pc = 0; pc < NPC; ++pc) load_b_piece(0, pc, wadr[0]);

#if defined(K16_CLOCK_PROBE)
  const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
  auto block = [&](auto itag, const int q0) {
#if K16_ROTATE_PRIO
    {  // issue arbitration is by priority, then age: the workgroups of the networks launched first ran ahead of their co-resident
       // partners (in-kernel span probe: 104 vs 125 us of a 119 us launch, 50 vs 67 us for conv2) and the younger ones finished
       // alone at half the pipe utilisation; rotating the priority every KS rows keeps the two waves of a SIMD level (conv_kyo.h)
      const int pr = ((int)blockIdx.y + q0 / KS) & 3;     // (every 2 KS / 4 KS rows, every row, every second row: all slower)
      if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
      else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
#endif
    // INTERIOR rows (round 4): KS steps in which every output row completed is an interior row of the band (row class P, its pooled
    // pair and the writer's pooled row exist, the next input row exists) -- the scalar bookkeeping of the general step (row class,
    // band / image limits, dropped-store offsets: ~45 of its ~80 SALU instructions and 6 of its 11 branches) folds away.
    constexpr bool IN = decltype(itag)::value;
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int q = q0 + sq;
      if (!IN && q >= qend) break;                   // uniform
      constexpr int PD_BASE = (KS - P) % KS;
      const int pdone = (PD_BASE + sq) % KS;
      f32x4 ctv[NT];                                 // f16 mode: E[row class of the output row this step completes][o][x = 0, 1, W - 2, W - 1]
      if (!B16) {
        const int yc = q - P;                        // (rows in front of a band are dropped below: any class will do)
        const int rc = IN ? P : (yc < P ? (yc < 0 ? 0 : yc) : (yc >= H - P ? 2 * P - (H - 1 - yc) : P));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int clo = pdone * NO - 16 * t, chi = pdone * NO + NO - 1 - 16 * t;
          if (chi >= 0 && clo <= 15) ctv[t] = lds_load<f32x4>(ctadr[t] + (uint32_t)(rc * NO * 16), 0);
        }
      }
      const int wy = (IN || q - P >= ymin) ? q - P : -1;     // the output row this step completes (rows in front of the band: none)
      const int wlo = ymin >> 1;                     // first pooled row of the band
      const int wpr = (IN || (wy >= 0 && (wy >> 1) - 1 >= wlo)) ? (wy >> 1) - 1 : -1;
      writer_load(wy & 1, wpr, IN);                  // (its LDS reads land under chunk 0's MFMAs)
      if (!IN && q >= H) writer_half(wy & 1, wpr);   // (the steps behind the image: no MFMAs to hide under)
      if (IN || q < H) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          if (ASYNC_A && !COUNTED_A) {
            // this chunk's operands were requested one row ago; issued since: the other chunks' loads and -- between chunk 0's
            // MFMAs and its loads -- the SH stores of the writer half every row carries (real or dropped: ONE wait count per chunk)
            if (ch == 0) k16_wait_vm<(NCH - 1) * LPC>(av[0][ch][0]);
            else k16_wait_vm<(NCH - 1) * LPC + SH>(av[0][ch][0]);
#pragma unroll
            for (int pa = 0; pa < NPA; ++pa)
#pragma unroll
              for (int m = 0; m < XT; ++m) {
                asm volatile("" : "+v"(av[pa][ch][m]));
                if (ODD) asm volatile("" : "+v"(ax[ch][m]));
              }
          }
          k16_u32x4 af[NPA][XT];
#pragma unroll
          for (int pa = 0; pa < NPA; ++pa)
#pragma unroll
          for (int m = 0; m < XT; ++m) {
            k16_u32x4 u = av[pa][ch][m];
            if (ODD) {
              const unsigned x4 = ax[ch][m];
              u = (k16_u32x4){__builtin_amdgcn_alignbyte(u[1], u[0], ashift), __builtin_amdgcn_alignbyte(u[2], u[1], ashift),
                              __builtin_amdgcn_alignbyte(u[3], u[2], ashift), __builtin_amdgcn_alignbyte(x4, u[3], ashift)};
            }
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (G::vgpr_may_need_mask(ch, m, v)) u[v] = (u[v] & emask[ch][m][v]) | ecst[ch][m][v];
            af[pa][m] = u;
          }
#pragma unroll
          for (int pc = NPC - 1; pc >= 0; --pc) {    // small pieces first; XT*NT independent accumulators between the pieces
#pragma unroll
            for (int pa = NPA - 1; pa >= 0; --pa) {
              if (B16 && pa + pc > B16) continue;
#pragma unroll
              for (int m = 0; m < XT; ++m)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                  if (B16) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(k16_bf16x8, af[pa][m]),
                                                                              __builtin_bit_cast(k16_bf16x8, bv[t][pc]), acc[m][t], 0, 0, 0);
                  else acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[pa][m]), bv[t][pc], acc[m][t], 0, 0, 0);
                }
            }
            // this piece's registers: the same piece of the next chunk (pinned here: the scheduler would sink the reads to the
            // end of the chunk and the next chunk would open waiting for them)
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 1 < NCH) load_b_piece(ch + 1, pc, wadr[sq]);
            else if (IN || q + 1 < H) load_b_piece(0, pc, wadr[(sq + 1) % KS]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (ch == 0) writer_half(wy & 1, wpr, IN);
          if (ASYNC_A || q + 1 < H) load_a(ch, q + 1);   // this chunk's operands of the next row, a whole row period ahead (ASYNC_A:
                                                         // also behind the last row -- masked by the descriptor, never used -- so
                                                         // that the hand-counted waits see the same sequence in every row)
          if (!ASYNC_A || COUNTED_A) __builtin_amdgcn_sched_barrier(0);      // (compiler-counted loads stay where they are issued: left alone, the scheduler sinks them to their first use)
        }
      }
      const int y = (IN || q - P >= ymin) ? q - P : -1;      // (rows in front of the band: partial sums, dropped like the rows above the image)
      const int par = y & 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int clo = pdone * NO - 16 * t, chi = pdone * NO + NO - 1 - 16 * t;
        if (chi >= 0 && clo <= 15) {
          const bool inr = li >= clo && li <= chi;
          if (inr) {
#pragma unroll
            for (int m = 0; m < XT; ++m) {
              // f16 mode: the border columns' data-independent remainder (file comment), requested from LDS before the row's MFMAs
              f32x4 zc = acc[m][t];
              if (!B16 && (IN || y >= 0)) {
                if (m == 0) { zc[0] = fmaf(-fl, ctv[t][0], zc[0]); zc[1] = fmaf(-fl, ctv[t][1], zc[1]); }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  zc[2 * h] = fmaf(-fr[m][h], ctv[t][2], zc[2 * h]); zc[2 * h + 1] = fmaf(-fr[m][h], ctv[t][3], zc[2 * h + 1]);
                }
              }
              if (PLAIN) {
                if (y >= 0) {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (sstrip * G::SW + m * 16 + 4 * lj + r < W)
                      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(zc[r] * inv), out_rsrc,
                                                            (int)eadr[t] + ((m * 16 + r) * NO) * 4, y * W * nout * 4, 0);
                }
              } else if (IN || y >= 0) {
                const uint32_t ea = eadr[t] + (uint32_t)((((y >> 1) & 1) * 2 + par) * (8 * XT * NO) * 8);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const float z0 = zc[2 * h], z1 = zc[2 * h + 1];
                  lds_store(ea, ((m * 8 + h) * NO) * 8, (f32x2){z1 > z0 ? z1 : z0, __int_as_float(z1 > z0 ? 1 : 0)});
                }
              }
              acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};
            }
          }
        }
      }
      // (a wave's LDS instructions execute in order: the writer half of a later step reads what this step's lanes stored without a
      // wait in between; the compiler must only keep the order)
      if (!PLAIN && (IN || y >= 0)) { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
    }
  };
  {
    auto interior = [&](const int q0) { return !PLAIN && q0 >= ymin + P + 2 && q0 >= 2 * P && q0 + KS < H && q0 + KS <= qend; };
    int q0 = qbeg;
    // (five chunks per row -- 30 channels -- sit at 254 VGPRs with the general step alone: the second copy of the loop spilled, 1125 -> 1173 us)
    if constexpr (NCH <= 3) {
      for (; q0 < qend && !interior(q0); q0 += KS) block(std::false_type{}, q0);      // the band's first rows
      for (; q0 < qend && interior(q0); q0 += KS) block(std::true_type{}, q0);
    }
    for (; q0 < qend; q0 += KS) block(std::false_type{}, q0);                       // ... and its last ones
  }
  {  // the last pair(s): the two steps behind the loop would have carried their writer halves
    const int ylast = qend - 1 - P;                  // last output row of this workgroup's band
#pragma unroll
    for (int k = 1; k <= 2; ++k) {
      const int wy = ylast + k, pr = (wy >> 1) - 1;
      // (half 0 of pair pr ran at step wy = 2 pr + 2, half 1 at 2 pr + 3: what has not run yet is wy > ylast; a pair exists if its second row does)
      const int prl = (pr >= (ymin >> 1) && 2 * pr + 1 <= ylast) ? pr : -1;
      if (ASYNC_A || prl >= 0) { writer_load(wy & 1, prl); writer_half(wy & 1, prl); }
    }
  }
  if (ASYNC_A && !COUNTED_A) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the look-ahead loads behind the last row)
  if (fuse3) {      // conv3 + pool3 of the workgroup's two images (all four waves are here: the launcher required B % IPW == 0)
    Conv3Ops ops;
    conv3_load_ops(ops, a.n3_w, a.n3_bias, nout, li, lj);      // (requested here: carried through the row loop they would cost 24 VGPRs)
    __syncthreads();
    const int im3 = wave >> 1;
    conv3_img_half(ops, img3 + im3 * C3_IMGF, wave & 1, a.n3_out + (long)(b0 + im3) * a.n3_out_bstride,
                   a.n3_amax + (long)(b0 + im3) * (C3_H / 2) * (C3_H / 2) * nout, nout, li, lj);
  }
#ifdef K16_CLOCK_PROBE
  if (lane == 0 && (blockIdx.x % 97) == 5 && blockIdx.y == 1 && wave == 0) {
    const unsigned long long pc1 = __builtin_readcyclecounter(), pr1 = __builtin_amdgcn_s_memrealtime();
    printf("K16CLK cin %d block %d: %llu core cycles, %llu ref ticks (100 MHz) in the row loop -> %.3f GHz; weight image %llu ticks, setup to loop %llu ticks\n", CIN, (int)blockIdx.x, pc1 - pc0, pr1 - pr0,
           (double)(pc1 - pc0) / (10.0 * (double)(pr1 - pr0)), pe1 - pe0, pr0 - pe1);
  }
#endif
}

template <int CIN, int KS, int XT, int IPW, bool PLAIN = false, int B16 = 0, int NPCS = F16_PIECES>      // B16: 0, or B16_SIX / B16_NINE
static inline int conv_fwd_k16_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef K16Geom<CIN, KS, XT, IPW, (B16 != 0), NPCS> G;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)G::LDS_BYTES;
  auto kern = conv_fwd_k16_kernel<CIN, KS, XT, IPW, PLAIN, B16, NPCS>;
  static bool attr_done[CPP_MAX_DEVICES] = {};          // (kernel attributes are per device: one cpp_ctx per GPU may share the process)
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int grid = ((a.B + IPW - 1) / IPW) * (a.nbands > 1 ? a.nbands : 1);
  hipLaunchKernelGGL(kern, dim3(grid, batch.n), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

// conv1 of f16 image batches with one whitening table for the batch (white_bstride == 0), even CIN and W, 5x5
int conv_fwd_k16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool plain, const ConvArgsN& a, bool* handled);
// conv1 of 64-pixel-wide 18-channel f16 image batches on conv_rs16.h (a.wimg: every network's operand image buffer)
int conv_fwd_rs16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, bool plain, const ConvArgsN& a, bool* handled);
size_t conv_rs16_image_bytes();
// conv2 forward from the bf16 planes conv1 left (a.in_b16): 10 channels, 5x5, even W <= 64
int conv_fwd_kb16_dispatch(cpp_ctx* ctx, int cin, int ks, int in_mode, const ConvArgsN& a, bool* handled);

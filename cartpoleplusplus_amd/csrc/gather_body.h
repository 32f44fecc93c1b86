// The fused uniform-sample + minibatch-gather + whitening-statistics pass as a device function (replay_memory.py:123-138 +
// base_network.py:95-96): gather_stats_kernel (replay.hip) is a wrapper around it, and so are the launches it shares with another
// kernel (reduce_gather_kernel, conv1_dw_gather_kernel).  The caller provides its LDS: sh [256 * GATHER_SH] floats, dsh
// [CPP_MAX_CHANNELS * 16] doubles, lut [256] floats.
//
// The sums are EXACT for pixel states (round 4): var = E[x^2] - mu^2 cancels 20-80 x on a render's near-constant channels (a sky, a
// floor), where the f32 per-lane chains of rounds 1-3 (36 terms each, 1e-7 relative) moved the whitening scale by 2e-6 relative --
// 5e-5 absolute on conv1 outputs of magnitude 13 (profiles/experiments/r04_render_probe.txt).  A lane now carries both sums in f64 (the square of an f16 has 22 bits, a
// lane adds <= ~10^2 of them: exact); they cross LDS as (hi, lo) float pairs and are combined in f64 as before.
#pragma once
#include "common.h"
constexpr int GATHER_SH = 16;      // floats per thread: (hi, lo) of one statistic's 8 element sums
constexpr int GATHER_LDS_BYTES = 256 * GATHER_SH * 4 + CPP_MAX_CHANNELS * 16 * 8 + 256 * 4;

__device__ __forceinline__ int sample_row(uint64_t seed, uint64_t counter, int b, int size) {
  u32x4 c = {(uint32_t)b, 0u, (uint32_t)counter, (uint32_t)(counter >> 32)};
  const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (int)(((uint64_t)r.x * (uint64_t)size) >> 32);      // uniform in [0, size)
}

template <typename T> struct Vec8;
template <> struct Vec8<__half> {
  typedef __half Out;
  uint4 raw;
  __device__ void load(const __half* p) { raw = *reinterpret_cast<const uint4*>(p); }
  __device__ void store(__half* p) const { *reinterpret_cast<uint4*>(p) = raw; }
  __device__ float get(int e) const {
    const uint32_t w = e < 2 ? raw.x : (e < 4 ? raw.y : (e < 6 ? raw.z : raw.w));
    const unsigned short h = (e & 1) ? (unsigned short)(w >> 16) : (unsigned short)(w & 0xffffu);
    return __half2float(__ushort_as_half(h));
  }
};
// 8-bit pixel codes (CPP_U8 store): 8 codes per vector, looked up in the f16(k/255) table (LDS copy); gathered as f16
template <> struct Vec8<uint8_t> {
  typedef __half Out;
  uint2 raw; const float* lut;
  __device__ void load(const uint8_t* p) { raw = *reinterpret_cast<const uint2*>(p); }
  __device__ float get(int e) const { return lut[((e < 4 ? raw.x : raw.y) >> (8 * (e & 3))) & 0xffu]; }
  __device__ void store(__half* p) const {
    uint4 o;
    uint32_t* w = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int e = 0; e < 8; e += 2)
      w[e >> 1] = (uint32_t)__half_as_ushort(__float2half(get(e))) | ((uint32_t)__half_as_ushort(__float2half(get(e + 1))) << 16);
    *reinterpret_cast<uint4*>(p) = o;
  }
};
template <> struct Vec8<float> {
  typedef float Out;
  float4 a, b;
  __device__ void load(const float* p) {
    a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = a; *reinterpret_cast<float4*>(p + 4) = b;
  }
  __device__ float get(int e) const {
    switch (e) { case 0: return a.x; case 1: return a.y; case 2: return a.z; case 3: return a.w;
                 case 4: return b.x; case 5: return b.y; case 6: return b.z; default: return b.w; }
  }
};

template <typename V> __device__ __forceinline__ void vec_init(V&, const float*) {}
__device__ __forceinline__ void vec_init(Vec8<uint8_t>& v, const float* lut) { v.lut = lut; }
template <typename T, typename O> __device__ __forceinline__ O elem_convert(T x, const float*) { return (O)x; }
template <> __device__ __forceinline__ __half elem_convert<uint8_t, __half>(uint8_t k, const float* lut) { return __float2half(lut[k]); }

__host__ __device__ inline int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

// (b, which): the workgroup's place in the (B, 2) grid -- blockIdx for a launch of its own
template <typename T>
__device__ __forceinline__ void gather_stats_body(const GatherArgs& a, const int b, const int which, float* sh, double* dsh, float* lut) {
  // (lut: CPP_U8 store, f16(k/255) as float)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr bool U8 = sizeof(T) == 1;
  if (U8) { lut[tid] = __half2float(a.lut[tid]); __syncthreads(); }
  typedef typename Vec8<T>::Out OutT;

  // --- sample + double indirection on lane 0, broadcast by wavefront shuffle
  int row = 0, slot = 0;
  if (lane == 0) {
    if (a.s_idx[0] == nullptr) { row = b; slot = b; }       // statistics over an already gathered batch
    else {
      row = a.rows ? a.rows[b] : sample_row(a.seed, (a.counter ? *a.counter : 0) + (uint64_t)a.counter_add, b, a.size_ptr ? *a.size_ptr : a.size);
      slot = a.s_idx[which][row];
    }
  }
  row = __shfl(row, 0);
  slot = __shfl(slot, 0);

  if (tid == 0 && a.out_slot[which]) a.out_slot[which][b] = slot;
  if (which == 0 && a.s_idx[0] != nullptr) {
    if (tid == 0 && a.rows_out) a.rows_out[b] = row;
    if (tid < a.action_dim) a.out_action[(long)b * a.action_dim + tid] = a.action[(long)row * a.action_dim + tid];
    if (tid == 64) a.out_reward[b] = a.reward[row];
    if (tid == 65) a.out_mask[b] = a.mask[row];
  }

  const T* src = (const T*)a.store[which] + (long)slot * a.elems;
  OutT* dst = a.out_state[which] ? (OutT*)a.out_state[which] + (long)b * a.elems : nullptr;
  const long nvec = a.elems >> 3;
  const int C = a.C;

  // The store already holds this state's sums (computed by this same function when the state was written: launch_slot_stats) and
  // nobody wants a copy of the pixels (conv1 reads the store through out_slot): the sampled row costs 2 C doubles, not the image.
  if (a.slot_stats != nullptr && dst == nullptr && C > 0) {
    if (tid < 2 * C) a.part[((long)which * a.B + b) * 2 * C + tid] = a.slot_stats[(long)slot * 2 * C + tid];
    return;
  }

  if (C <= 0) {                                   // gather only (low-dim states)
    for (long v = tid; v < nvec; v += 256) { Vec8<T> x; vec_init(x, lut); x.load(src + v * 8); if (dst) x.store(dst + v * 8); }
    for (long e = nvec * 8 + tid; e < a.elems; e += 256) if (dst) dst[e] = elem_convert<T, OutT>(src[e], lut);
    return;
  }

  const int P = C / gcd_int(8, C);
  const int act = (64 / P) * P;                   // lanes in use per wave: multiple of the period
  double s[8], ss[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.0; ss[e] = 0.0; }
  if (lane < act) {
    // GU row vectors per thread in flight (one 16-byte load each is far too little to cover the HBM latency with two
    // workgroups per CU); the accumulation order per lane is unchanged
    constexpr int GU = 6;
    long v = wave * act + lane;
    const long stride = 4 * act;
    for (; v + (GU - 1) * stride < nvec; v += GU * stride) {
      Vec8<T> x[GU];
#pragma unroll
      for (int u = 0; u < GU; ++u) { vec_init(x[u], lut); x[u].load(src + (v + u * stride) * 8); }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        if (dst) x[u].store(dst + (v + u * stride) * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const double f = (double)x[u].get(e); s[e] += f; ss[e] = fma(f, f, ss[e]); }
      }
    }
    for (; v < nvec; v += stride) {
      Vec8<T> x;
      vec_init(x, lut);
      x.load(src + v * 8);
      if (dst) x.store(dst + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const double f = (double)x.get(e); s[e] += f; ss[e] = fma(f, f, ss[e]); }
    }
  }
  // stage 2: per (class q, element e): sum the lanes of that class over the 4 waves, in f64.  One statistic at a time, each f64
  // lane sum crossing LDS as a (hi, lo) float pair (48 bits of a sum of <= ~10^2 terms of <= 22 bits: nothing is lost)
#pragma unroll
  for (int stat = 0; stat < 2; ++stat) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double v = stat ? ss[e] : s[e];
      const float hi = (float)v;
      sh[tid * GATHER_SH + e] = hi; sh[tid * GATHER_SH + 8 + e] = (float)(v - (double)hi);
    }
    __syncthreads();
    if (tid < P * 8) {
      const int q = tid >> 3, e = tid & 7;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;    // one chain per wave: four LDS reads in flight instead of one
      for (int l = q; l < act; l += P) {
        a0 += (double)sh[(0 * 64 + l) * GATHER_SH + e] + (double)sh[(0 * 64 + l) * GATHER_SH + 8 + e];
        a1 += (double)sh[(1 * 64 + l) * GATHER_SH + e] + (double)sh[(1 * 64 + l) * GATHER_SH + 8 + e];
        a2 += (double)sh[(2 * 64 + l) * GATHER_SH + e] + (double)sh[(2 * 64 + l) * GATHER_SH + 8 + e];
        a3 += (double)sh[(3 * 64 + l) * GATHER_SH + e] + (double)sh[(3 * 64 + l) * GATHER_SH + 8 + e];
      }
      dsh[q * 16 + stat * 8 + e] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
  }
  // stage 3: per channel: the (q, e) pairs with (8q + e) % C == c
  if (tid < 2 * C) {
    const int stat = tid / C, c = tid - stat * C;
    double acc = 0.0;
    for (int q = 0; q < P; ++q) {                  // elements e of class q with (8 q + e) % C == c, ascending
      int e = (c - 8 * q) % C; if (e < 0) e += C;
      for (; e < 8; e += C) acc += dsh[q * 16 + stat * 8 + e];
    }
    a.part[((long)which * a.B + b) * 2 * C + tid] = acc;
  }
}


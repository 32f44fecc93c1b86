// Device-resident replay memory kernels: fused uniform-sample + minibatch-gather + whitening
// statistics (replay_memory.py:123-138 + base_network.py:95-96), plus helpers.
//
// gather_stats_kernel: grid (B, 2) -- one 256-thread workgroup per sampled row and per state column
// (state_1 / state_2).  Lane 0 of each wave draws the row (Philox4x32-10 keyed by (seed, counter), or
// the caller's index), follows the double indirection state[state_k_idx[row]] and broadcasts the slot
// with a wavefront shuffle; all lanes then stream the 144 KiB f16 state with 16-byte loads/stores.
// Per-channel sum(x), sum(x^2) are accumulated in the same pass: a lane only ever touches vectors
// v = lane (mod P), P = C / gcd(8, C), so the 8 elements of every vector it loads belong to the same 8
// channels and the accumulators are plain registers.  Row partials leave the kernel in f64 and are
// combined in fixed order by stats_finalize_kernel (deterministic).
#include <cstring>
#include "conv_impl.h"
#include "gather_body.h"
#include "stats_body.h"

template <typename T>
__global__ __launch_bounds__(256) void gather_stats_kernel(const GatherArgs a) {
  __shared__ float sh[256 * GATHER_SH];
  __shared__ double dsh[CPP_MAX_CHANNELS * 16];
  __shared__ float lut[256];
  gather_stats_body<T>(a, (int)blockIdx.x, (int)blockIdx.y, sh, dsh, lut);
}

// The dW reductions that end a minibatch's backward pass and the sample + statistics pass that starts the next minibatch in ONE
// launch: the gather depends on nothing the step computes (its draw is keyed by the sampler's counter + 1), the reduction
// is a latency chain on a handful of workgroups -- back to back they cost 8.8 + 19.8 us, together the longer of the two.
// Workgroups [0, reduction blocks) reduce (4 slices of 64 lanes: its own fixed order, the same for every store type), the rest gather.
template <typename T>
__global__ __launch_bounds__(256) void reduce_gather_kernel(const DwReduceBatch rb, const GatherArgs a) {
  const int nred = rb.block_start[rb.n];
  if ((int)blockIdx.x < nred) {
    __shared__ float red[4][64];
    conv_dw_reduce_body<4>(rb, (int)blockIdx.x, red);
  } else {
    __shared__ float sh[256 * GATHER_SH];
    __shared__ double dsh[CPP_MAX_CHANNELS * 16];
    __shared__ float lut[256];
    const int i = (int)blockIdx.x - nred;
    gather_stats_body<T>(a, i % a.B, i / a.B, sh, dsh, lut);
  }
}

int launch_reduce_gather(cpp_ctx* ctx, const DwReduceBatch& rb, const GatherArgs& a, int dtype) {
  prof_begin(ctx);
  const dim3 grid(rb.block_start[rb.n] + 2 * a.B);
  if (dtype == 2) hipLaunchKernelGGL(reduce_gather_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, rb, a);
  else hipLaunchKernelGGL(reduce_gather_kernel<__half>, grid, dim3(256), 0, ctx->stream, rb, a);
  LAUNCH_CHECK();
  prof_end(ctx, K_REDUCE_GATHER);
  return 0;
}

// One workgroup per state: gather_stats_body in its "statistics over rows that are already in place" mode (s_idx == nullptr: row b of
// the store is state b), so a state's stored sums are bit for bit what a gather of that state computes.
template <typename T>
__global__ __launch_bounds__(256) void slot_stats_kernel(const GatherArgs a, const int32_t* slots, int first) {
  __shared__ float sh[256 * GATHER_SH];
  __shared__ double dsh[CPP_MAX_CHANNELS * 16];
  __shared__ float lut[256];
  const int slot = slots ? slots[blockIdx.x] : first + (int)blockIdx.x;
  gather_stats_body<T>(a, slot, 0, sh, dsh, lut);
}

int launch_slot_stats(cpp_ctx* ctx, const void* store, int dtype, long elems, int C, double* slot_stats, const int32_t* slots, int first, int n,
                      const __half* lut) {
  if (n <= 0) return 0;
  GatherArgs a; memset(&a, 0, sizeof(a));
  a.store[0] = store; a.part = slot_stats; a.elems = elems; a.C = C; a.B = 0; a.lut = lut;      // (which = 0: part index = slot * 2C + tid)
  prof_begin(ctx);
  if (dtype == 2) hipLaunchKernelGGL(slot_stats_kernel<uint8_t>, dim3(n), dim3(256), 0, ctx->stream, a, slots, first);
  else hipLaunchKernelGGL(slot_stats_kernel<__half>, dim3(n), dim3(256), 0, ctx->stream, a, slots, first);
  LAUNCH_CHECK();
  prof_end(ctx, K_GATHER_STATS);
  return 0;
}

int launch_gather_stats(cpp_ctx* ctx, const GatherArgs& a, int dtype) {
  if (a.C > CPP_MAX_CHANNELS) { cpp_set_error("gather: %d channels > %d", a.C, CPP_MAX_CHANNELS); return 1; }
  if (a.C > 0 && (a.elems % 8 != 0 || a.elems % a.C != 0 || a.C / gcd_int(8, a.C) > 16)) {
    cpp_set_error("gather: vector statistics path needs state_elems %ld %% 8 == 0 and period <= 16 (C=%d)",
                  a.elems, a.C);
    return 1;
  }
  prof_begin(ctx);
  if (dtype == 2) hipLaunchKernelGGL(gather_stats_kernel<uint8_t>, dim3(a.B, 2), dim3(256), 0, ctx->stream, a);
  else if (dtype == 1) hipLaunchKernelGGL(gather_stats_kernel<__half>, dim3(a.B, 2), dim3(256), 0, ctx->stream, a);
  else hipLaunchKernelGGL(gather_stats_kernel<float>, dim3(a.B, 2), dim3(256), 0, ctx->stream, a);
  LAUNCH_CHECK();
  prof_end(ctx, K_GATHER_STATS);
  return 0;
}

// white[w][0][c] = rsqrt(var + 1e-6), white[w][1][c] = -mean * rsqrt(var + 1e-6)   (base_network.py:97-99)
__global__ __launch_bounds__(64) void stats_finalize_kernel(const double* part, int nparts, int which_count,
                                                            int C, double count, float* white, double eps, uint64_t* bump, unsigned* wmax) {
  if (bump && blockIdx.x == 0 && threadIdx.x == 0) *bump += 1;
  stats_finalize_wave(part, nparts, C, count, white, eps, (int)blockIdx.x, (int)threadIdx.x, wmax);
}

int launch_stats_finalize(cpp_ctx* ctx, const double* part, int nparts, int which_count, int C,
                          double count, float* white, double eps, uint64_t* bump, unsigned* wmax) {
  prof_begin(ctx);
  hipLaunchKernelGGL(stats_finalize_kernel, dim3(which_count * C), dim3(64), 0, ctx->stream, part, nparts,
                     which_count, C, count, white, eps, bump, wmax);
  LAUNCH_CHECK();
  prof_end(ctx, K_STATS_FINALIZE);
  return 0;
}

// fallback for shapes the vector path cannot take (elems % 8 != 0): one workgroup per channel
template <typename T>
__global__ __launch_bounds__(256) void stats_generic_kernel(const T* x, long npix, int C, float* white, double eps) {
  __shared__ double r0[256], r1[256];
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (long p = threadIdx.x; p < npix; p += 256) {
    const double f = (double)(float)x[p * C + c];
    s += f; ss += f * f;
  }
  r0[threadIdx.x] = s; r1[threadIdx.x] = ss;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) white_from_moments(r0[0], r1[0], (double)npix, eps, &white[c], &white[C + c]);
}

int launch_stats_generic(cpp_ctx* ctx, const void* x, int dtype, long npix, int C, float* white, double eps) {
  prof_begin(ctx);
  if (dtype == 1) hipLaunchKernelGGL(stats_generic_kernel<__half>, dim3(C), dim3(256), 0, ctx->stream,
                                     (const __half*)x, npix, C, white, eps);
  else hipLaunchKernelGGL(stats_generic_kernel<float>, dim3(C), dim3(256), 0, ctx->stream,
                          (const float*)x, npix, C, white, eps);
  LAUNCH_CHECK();
  prof_end(ctx, K_STATS_GENERIC);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// synthetic fill (bench / tests): SURVEY 8d inputs generated on the device
// ---------------------------------------------------------------------------------------------
// (grid-stride loops: a store of tens of GB has more 16-element blocks than one launch may have threads -- 2^32)
constexpr unsigned FILL_GRID = 1u << 20;
__global__ void replay_fill_states_kernel(__half* store, long total, uint64_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i * 16 < total; i += (long)gridDim.x * blockDim.x) {   // 16 elements per step
    u32x4 c = {(uint32_t)i, (uint32_t)(i >> 32), 0x5eedu, 1u};
    const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 16; ++e) {
      const long idx = i * 16 + e;
      if (idx < total) {
        const uint32_t k = (w[e >> 2] >> (8 * (e & 3))) & 0xffu;
        store[idx] = __float2half((float)k / 255.0f);             // f16(k/255): bullet_cartpole.py:239-243
      }
    }
  }
}

__global__ void replay_fill_rows_kernel(int32_t* s1, int32_t* s2, float* action, float* reward,
                                        float* mask, int rows, int action_dim, uint64_t seed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const int ep = i / 50;                         // fixed 50-step episodes: terminal w.p. 1/50
  s1[i] = i + ep;
  s2[i] = i + ep + 1;
  reward[i] = 1.0f;
  mask[i] = (i % 50 == 49) ? 0.0f : 1.0f;
  u32x4 c = {(uint32_t)i, 0u, 0xac7u, 2u};
  const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
  for (int k = 0; k < action_dim; ++k)
    action[(long)i * action_dim + k] = (float)(w[k & 3] >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

int launch_replay_fill(cpp_ctx* ctx, __half* store, long elems, int slots, int32_t* s1, int32_t* s2,
                       float* action, float* reward, float* mask, int rows, int action_dim,
                       uint64_t seed) {
  const long total = elems * (long)slots;
  const long nthreads = (total + 15) / 16;
  prof_begin(ctx);
  if (store) {                                       // (nullptr: the caller fills a CPP_U8 store itself)
    const long nblk = (nthreads + 255) / 256;
    hipLaunchKernelGGL(replay_fill_states_kernel, dim3(nblk < (long)FILL_GRID ? (unsigned)nblk : FILL_GRID), dim3(256), 0,
                       ctx->stream, store, total, seed);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(replay_fill_rows_kernel, dim3((rows + 255) / 256), dim3(256), 0, ctx->stream, s1, s2,
                     action, reward, mask, rows, action_dim, seed);
  LAUNCH_CHECK();
  prof_end(ctx, K_REPLAY_FILL);
  return 0;
}

__global__ void f32_to_f16_kernel(__half* dst, const float* src, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = __float2half(src[i]);      // round-to-nearest-even, like numpy's astype(float16)
}

int launch_f32_to_f16(cpp_ctx* ctx, __half* dst, const float* src, long n) {
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     dst, src, n);
  LAUNCH_CHECK();
  return 0;
}

// CPP_U8 store: states must be 8-bit pixel images, f16(x) == f16(k/255) for some code k (bullet_cartpole.py:239-243); the
// store keeps k.  Anything else raises the `bad` flag (the caller reports an error instead of storing a lossy copy).
template <typename T>
__global__ void to_u8_kernel(uint8_t* dst, const T* src, long n, const __half* lut, int* bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const __half h = __float2half((float)src[i]);          // what the f16 store would hold (RNE, like numpy)
  int k = (int)rintf(__half2float(h) * 255.0f);
  k = k < 0 ? 0 : (k > 255 ? 255 : k);
  if (__half_as_ushort(lut[k]) != __half_as_ushort(h)) atomicOr(bad, 1);
  dst[i] = (uint8_t)k;
}

int launch_to_u8(cpp_ctx* ctx, uint8_t* dst, const void* src, int src_dtype, long n, const __half* lut, int* bad) {
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (src_dtype == 1) hipLaunchKernelGGL(to_u8_kernel<__half>, dim3(grid), dim3(256), 0, ctx->stream, dst, (const __half*)src, n, lut, bad);
  else hipLaunchKernelGGL(to_u8_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, dst, (const float*)src, n, lut, bad);
  LAUNCH_CHECK();
  return 0;
}

// raw camera bytes into an f16 store: dst = f16(k/255) through the table (the reference's render conversion)
__global__ void u8_to_f16_kernel(__half* dst, const uint8_t* src, long n, const __half* lut) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = lut[src[i]];
}

int launch_u8_to_f16(cpp_ctx* ctx, __half* dst, const uint8_t* src, long n, const __half* lut) {
  hipLaunchKernelGGL(u8_to_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst, src, n, lut);
  LAUNCH_CHECK();
  return 0;
}

__global__ void replay_fill_u8_kernel(uint8_t* store, long total, uint64_t seed) {      // same codes as replay_fill_states_kernel
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i * 16 < total; i += (long)gridDim.x * blockDim.x) {
    u32x4 c = {(uint32_t)i, (uint32_t)(i >> 32), 0x5eedu, 1u};
    const u32x4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 16; ++e) {
      const long idx = i * 16 + e;
      if (idx < total) store[idx] = (uint8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
    }
  }
}

int launch_replay_fill_u8(cpp_ctx* ctx, uint8_t* store, long total, uint64_t seed) {
  const long nthreads = (total + 15) / 16;
  const long nblk = (nthreads + 255) / 256;
  hipLaunchKernelGGL(replay_fill_u8_kernel, dim3(nblk < (long)FILL_GRID ? (unsigned)nblk : FILL_GRID), dim3(256), 0, ctx->stream, store, total, seed);
  LAUNCH_CHECK();
  return 0;
}

// (unless: non-null and set -> the counter stays: an optimiser step that stands down is not counted)
__global__ void counter_add_kernel(uint64_t* counter, uint64_t inc, const int* unless) { if (!unless || !*unless) *counter += inc; }

int launch_counter_add(cpp_ctx* ctx, uint64_t* counter, uint64_t inc, const int* unless) {
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, ctx->stream, counter, inc, unless);
  LAUNCH_CHECK();
  return 0;
}

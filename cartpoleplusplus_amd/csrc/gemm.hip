// Small fused GEMM (+activation / activation-gradient epilogue) on v_mfma_f32_16x16x4_f32 for the
// actor / critic MLP heads (base_network.py:58-71, ddpg_cartpole.py:95-100, :168-171, :180-184) and a
// few elementwise helpers.  Biases ride along as the last row of each weight matrix: the flat parameter
// layout stores "<scope>/biases" directly after "<scope>/weights", so [W; b] is one (n_in+1, n_out)
// matrix and every activation buffer carries a constant 1.0 in its last column -- forward bias add,
// db = sum(dz) and dW = x^T dz all fall out of the same GEMM.
#include "common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#if GEMM_SUB_MIN_K < 100000
#define GEMM_RED (4 * 256)      // floats per wave in the cross-wave reduction: 2 x 2 sub-tiles
#else
#define GEMM_RED 256
#endif

// One 4-wave workgroup per 16x16 output tile, K split across the waves in interleaved chunks of 64 (the
// matrices are tiny and L2 resident, so the kernel is a latency chain: 32 loads in flight per lane and a
// 4x shorter chain matter more than tiling).  Partial tiles are combined through LDS in fixed order.
// A(m,k) = A[m*sAm + k*sAk], B(k,n) = B[k*sBk + n*sBn]: the strides express the transposes of the
// backward passes (dX = dz W^T, dW = x^T dz) without materialising them.
#ifndef GEMM_R
#define GEMM_R 1      // rounds of a wave whose loads are in flight together.  Measured (six MLP levels of a step, round 1's branchy loads): 1 -> 0.0440 ms, 2 -> 0.0439, 4 -> 0.0457 (120 VGPRs), 6 -> 0.0525 (168), 11 -> 0.1112 (290: one workgroup per CU); with the branch-free loads below (four levels): 1 -> 0.0431, 2 -> 0.0430, 3 -> 0.0437, 5 -> 0.0452 (previous loads on the same box: 0.0441) -- the levels are not a chain of round trips
#endif
#ifndef GEMM_U
#define GEMM_U 4      // k values per wave and round = 4 U (sweep 4 / 8 / 12 / 16 with 16-byte loads: 0.057 / 0.058 / 0.064 / 0.064 ms for the six MLP levels)
#endif
// S x S sub-tiles of 16 x 16 per workgroup (gemm_sub(): always 1 as shipped).  With one tile per workgroup every k value of an
// operand row is fetched once per tile that needs it: the K = 641 level moves 52 MB through L1 for 2 MB of matrices and runs at
// 0.87 us per round in every workgroup at once (in-kernel probe, -DGEMM_CLOCK).  2 x 2 sub-tiles share each operand fragment
// between two MFMAs and halve that traffic -- and are SLOWER (0.0565 vs 0.0433 ms for the four levels; -DGEMM_SUB_MIN_K=256): a
// quarter of the workgroups, each four times as long, on a chip the 1 x 1 tiling does not fill either.  Kept for the record and for
// larger problems.  Every output sums its products in the same order in both (same K split over the waves, same k order inside a
// round): bit-identical results.
template <bool VA, bool VB, int S>
__device__ __forceinline__ void gemm_tile_t(const GemmArgs& g, int tile, float (*red)[GEMM_RED]) {
#ifdef GEMM_CLOCK
  const unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int tiles_n = (g.N + 16 * S - 1) / (16 * S);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  int mrow[S], ncol[S]; bool mv[S], nv[S];
#pragma unroll
  for (int i = 0; i < S; ++i) {
    mrow[i] = (tm * S + i) * 16 + li; mv[i] = mrow[i] < g.M;
    ncol[i] = (tn * S + i) * 16 + li; nv[i] = ncol[i] < g.N;
  }
  f32x4 acc[S][S];
#pragma unroll
  for (int i = 0; i < S; ++i)
#pragma unroll
    for (int j = 0; j < S; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = GEMM_U;
  // k order inside a round of 4 U values: lane group lj takes k0 + U lj + u (u = MFMA step), so an operand that is contiguous
  // in k (activations / dz in the forward and dX passes: sAk == 1; W^T in dX: sBk == 1) is read with 16-byte loads --
  // a quarter of the load instructions and of the cache lines the texture path walks.  Any k order works as long as A and B agree.
  static_assert(U % 4 == 0, "whole float4s per lane");
  constexpr bool va = VA, vb = VB;                   // (compile-time: one straight-line loop body per combination; gemm_tile below)
  // No branch around any load: an operand is read through a buffer descriptor that ends with its last element, and an
  // element outside the matrix (row >= M, column >= N, k >= K) is requested at an offset behind that end -- it comes back as
  // zero.  (With `in range ? load : 0` every load sat in a basic block of its own and the zero of the other path had to wait
  // for the load before it could overwrite the register: two dependent round trips per round.)  A 16-byte load that straddles
  // k = K is in range as a whole; its tail is cleared by selects.
  typedef unsigned gemm_u4 __attribute__((ext_vector_type(4)));
  constexpr int OOB = 0x7FFFFF00;
  constexpr int AUX = 0;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.A), 0, (int)((((long)(g.M - 1) * g.sAm + (long)(g.K - 1) * g.sAk) + 1) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(g.B), 0, (int)((((long)(g.K - 1) * g.sBk + (long)(g.N - 1) * g.sBn) + 1) * 4), 0x00020000);
  int abase[S], bbase[S];
#pragma unroll
  for (int i = 0; i < S; ++i) { abase[i] = (int)((long)mrow[i] * g.sAm * 4); bbase[i] = (int)((long)ncol[i] * g.sBn * 4); }
  const int ask = (int)(g.sAk * 4), bsk = (int)(g.sBk * 4);
  // the epilogue's own operands (the forward activation of a ReLU / tanh gradient, the running C of an accumulating GEMM) are requested
  // HERE, by the wave that will use them: behind the reduction they were one more round trip to the memory side per level
  float ypre[S][S][4], cpre[S][S][4];
  const bool want_y = g.Y != nullptr && (g.epi == GE_MUL_RELU_GRAD || g.epi == GE_MUL_RELU_GRAD_X2 || g.epi == GE_MUL_TANH_GRAD || g.epi == GE_ACTOR_HEAD);
#pragma unroll
  for (int ti = 0; ti < S; ++ti)
#pragma unroll
    for (int tj = 0; tj < S; ++tj)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tm * S + ti) * 16 + 4 * lj + i, n = ncol[tj];
        const bool on = wave == 0 && nv[tj] && row < g.M;
        ypre[ti][tj][i] = (on && want_y) ? g.Y[(long)row * g.ldy + n] : 0.f;
        cpre[ti][tj][i] = (on && g.accumulate) ? g.C[(long)row * g.ldc + n] : 0.f;
      }
  // A wave's rounds are independent until the MFMAs: ALL operand loads of up to GEMM_R rounds are issued before the first
  // one is used.  Same k order, same summation order as a plain loop.
  constexpr int R = GEMM_R;
  for (int kr = wave * 4 * U; kr < g.K; kr += R * 4 * 4 * U) {
    gemm_u4 av[R][S][U / 4], bv[R][S][U / 4];        // (16-byte register tuples on both paths: no copies behind the loads)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int kb = kr + r * 4 * 4 * U + U * lj;
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
      for (int q = 0; q < U / 4; ++q) {
        const int k = kb + 4 * q;
        if (va) av[r][i][q] = __builtin_amdgcn_raw_buffer_load_b128(ra, (mv[i] & (k < g.K)) ? abase[i] + k * 4 : OOB, 0, AUX);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) av[r][i][q][e] = __builtin_amdgcn_raw_buffer_load_b32(ra, (mv[i] & (k + e < g.K)) ? abase[i] + (k + e) * ask : OOB, 0, AUX);
        }
        if (vb) bv[r][i][q] = __builtin_amdgcn_raw_buffer_load_b128(rb, (nv[i] & (k < g.K)) ? bbase[i] + k * 4 : OOB, 0, AUX);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[r][i][q][e] = __builtin_amdgcn_raw_buffer_load_b32(rb, (nv[i] & (k + e < g.K)) ? bbase[i] + (k + e) * bsk : OOB, 0, AUX);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);               // (the scheduler would pull the selects below up to their loads)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (kr + r * 4 * 4 * U < g.K) {                  // (uniform; rounds past K hold zeros anyway)
        const int kb = kr + r * 4 * 4 * U + U * lj;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // (behind ALL loads of the group: a select placed next to its load would wait for it before the next load is issued)
          float x[S], y[S];
#pragma unroll
          for (int i = 0; i < S; ++i) {
            x[i] = (va && kb + u >= g.K) ? 0.f : __uint_as_float(av[r][i][u >> 2][u & 3]);
            y[i] = (vb && kb + u >= g.K) ? 0.f : __uint_as_float(bv[r][i][u >> 2][u & 3]);
          }
#pragma unroll
          for (int i = 0; i < S; ++i)
#pragma unroll
            for (int j = 0; j < S; ++j) acc[i][j] = MFMA16(x[i], y[j], acc[i][j]);
        }
      }
    }
  }
#ifdef GEMM_CLOCK
  const unsigned long long c1 = __builtin_amdgcn_s_memrealtime();
#endif
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < S; ++i)
#pragma unroll
      for (int j = 0; j < S; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave - 1][(i * S + j) * 256 + lane * 4 + e] = acc[i][j][e];
  }
  __syncthreads();
#ifdef GEMM_CLOCK
  const unsigned long long c2 = __builtin_amdgcn_s_memrealtime();
#endif
  if (wave != 0) return;
  double sq = 0.0;
#pragma unroll
  for (int ti = 0; ti < S; ++ti)
#pragma unroll
  for (int tj = 0; tj < S; ++tj)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tm * S + ti) * 16 + 4 * lj + i, n = ncol[tj];
    if (nv[tj] && row < g.M) {
      const int ri = (ti * S + tj) * 256 + lane * 4 + i;
      float v = ((acc[ti][tj][i] + red[0][ri]) + red[1][ri]) + red[2][ri];
      if (g.accumulate) v += cpre[ti][tj][i];
      if (g.epi == GE_RELU) v = fmaxf(v, 0.f);
      else if (g.epi == GE_TANH) v = tanhf(v);
      else if (g.epi == GE_MUL_RELU_GRAD) v = ypre[ti][tj][i] > 0.f ? v : 0.f;
      else if (g.epi == GE_MUL_RELU_GRAD_X2) v = ypre[ti][tj][i] > 0.f ? 2.f * v : 0.f;
      else if (g.epi == GE_RELU_DROPOUT) {
        v = fmaxf(v, 0.f);
        if (g.drop_counter) {
          const uint64_t ctr = *g.drop_counter;
          const u32x4 c = {(uint32_t)(row * g.N + n), g.drop_layer, (uint32_t)ctr, (uint32_t)(ctr >> 32)};
          v = (philox4x32_10(c, g.drop_seed, 0u).x & 1u) ? 2.f * v : 0.f;
        }
      }
      else if (g.epi == GE_MUL_TANH_GRAD) { const float y = ypre[ti][tj][i]; v = v * (1.f - y * y); }
      g.C[(long)row * g.ldc + n] = v;
      sq += (double)v * (double)v;
      if (g.epi == GE_ACTOR_HEAD) { const float y = ypre[ti][tj][i]; g.C2[(long)row * g.ldc2 + n] = -v * (1.f - y * y); }
      else if (g.C2) g.C2[(long)row * g.ldc2 + n] = v;
    }
  }
#ifdef GEMM_CLOCK
  if (lane == 0 && g.K > 600 && (tile % 37) == 5)
    printf("GEMMCLK M %d N %d K %d tile %d (blk %d): loop %.2f us, reduce+barrier %.2f, epilogue %.2f (start tick %llu)\n", g.M, g.N, g.K, tile, (int)blockIdx.x,
           (c1 - c0) / 100.0, (c2 - c1) / 100.0, (__builtin_amdgcn_s_memrealtime() - c2) / 100.0, c0 % 100000000ull);
#endif
  if (g.sq_part) {                                   // (uniform) this workgroup's share of the gradient list's squared norm, fixed order
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) g.sq_part[tile] = sq;
  }
}

// an operand that is contiguous in k is read with 16-byte loads; long-K problems take 2 x 2 sub-tiles (uniform per problem)
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, int tile, float (*red)[GEMM_RED]) {
  const bool va = g.sAk == 1, vb = g.sBk == 1;
#if GEMM_SUB_MIN_K < 100000      // (not instantiated in the shipped build: gemm_sub() is 1 for every problem)
  if (gemm_sub(g.M, g.N, g.K) == 2) {
    if (va && vb) gemm_tile_t<true, true, 2>(g, tile, red);
    else if (va) gemm_tile_t<true, false, 2>(g, tile, red);
    else if (vb) gemm_tile_t<false, true, 2>(g, tile, red);
    else gemm_tile_t<false, false, 2>(g, tile, red);
    return;
  }
#endif
  if (va && vb) gemm_tile_t<true, true, 1>(g, tile, red);
  else if (va) gemm_tile_t<true, false, 1>(g, tile, red);
  else if (vb) gemm_tile_t<false, true, 1>(g, tile, red);
  else gemm_tile_t<false, false, 1>(g, tile, red);
}

// Workgroup -> tile, XCD-aware: workgroup b of a launch runs on XCD b mod 8 (round-robin dispatch), each XCD has its own L2, and
// both operands were written by earlier kernels, so every XCD fills its L2 with whatever its workgroups touch.  With tile = local
// index every XCD touches every row of A and every column of B; here the workgroups of one XCD take CONSECUTIVE tiles (row-major
// over (tm, tn)): an eighth of A's rows and all of B.  `first` = the launch-wide index of the problem's first workgroup.
// (Measured: 0.0431 vs 0.0439 ms for the four levels -- the fills are not what bounds a level either.)
#ifndef GEMM_XCD
#define GEMM_XCD 8
#endif
__device__ __forceinline__ int gemm_xcd_tile(int b, int first, int T) {
  if (GEMM_XCD <= 1) return b - first;
  const int lb = b - first, x = b % GEMM_XCD;
  // local workgroups on XCD x: lb = o, o + 8, ... with o = (x - first) mod 8; XCDs are ranked by o so that the ranks' tile ranges tile [0, T)
  const int o = ((x - first) % GEMM_XCD + GEMM_XCD) % GEMM_XCD;
  const int j = (lb - o) / GEMM_XCD;                          // this workgroup's position among its XCD's
  const int q = T / GEMM_XCD, rem = T % GEMM_XCD;             // rank o owns q (+ 1 if o < rem) local workgroups
  return o * q + (o < rem ? o : rem) + j;
}

__global__ __launch_bounds__(256) void gemm_mfma_kernel(const GemmArgs g) {
  __shared__ float red[3][GEMM_RED];
  gemm_tile(g, gemm_xcd_tile(blockIdx.x, 0, gridDim.x), red);
}

// several independent GEMMs in one launch: workgroup -> (problem, tile) through a prefix table
__global__ __launch_bounds__(256) void gemm_batch_kernel(const GemmBatch gb) {
  __shared__ float red[3][GEMM_RED];
  int p = 0;
  while (p + 1 < gb.n && (int)blockIdx.x >= gb.tile_start[p + 1]) ++p;
  gemm_tile(gb.g[p], gemm_xcd_tile(blockIdx.x, gb.tile_start[p], gb.tile_start[p + 1] - gb.tile_start[p]), red);
}

int launch_gemm(cpp_ctx* ctx, const GemmArgs& g) {
  const int tiles = gemm_tiles(g.M, g.N, g.K);
  prof_begin(ctx);
  hipLaunchKernelGGL(gemm_mfma_kernel, dim3(tiles), dim3(256), 0, ctx->stream, g);
  LAUNCH_CHECK();
  prof_end(ctx, K_GEMM);
  return 0;
}

int launch_gemm_batch(cpp_ctx* ctx, const GemmArgs* list, int n) {
  for (int i0 = 0; i0 < n; i0 += GEMM_BATCH_MAX) {
    const int cnt = n - i0 < GEMM_BATCH_MAX ? n - i0 : GEMM_BATCH_MAX;
    if (cnt == 1) { int rc = launch_gemm(ctx, list[i0]); if (rc) return rc; continue; }
    GemmBatch gb;
    gb.n = cnt; gb.tile_start[0] = 0;
    for (int i = 0; i < cnt; ++i) {
      gb.g[i] = list[i0 + i];
      gb.tile_start[i + 1] = gb.tile_start[i] + gemm_tiles(gb.g[i].M, gb.g[i].N, gb.g[i].K);
    }
    prof_begin(ctx);
    hipLaunchKernelGGL(gemm_batch_kernel, dim3(gb.tile_start[cnt]), dim3(256), 0, ctx->stream, gb);
    LAUNCH_CHECK();
    prof_end(ctx, K_GEMM);
  }
  return 0;
}

__global__ void copy_cols_kernel(float* dst, long ldd, int dcol0, const float* src, long lds_,
                                 int scol0, int ncols, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ncols) return;
  const int r = i / ncols, c = i - r * ncols;
  dst[(long)r * ldd + dcol0 + c] = src[(long)r * lds_ + scol0 + c];
}

int launch_copy_cols(cpp_ctx* ctx, float* dst, long ldd, int dcol0, const float* src, long lds_,
                     int scol0, int ncols, int rows) {
  const int n = rows * ncols;
  if (n <= 0) return 0;
  prof_begin(ctx);
  hipLaunchKernelGGL(copy_cols_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dst, ldd,
                     dcol0, src, lds_, scol0, ncols, rows);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

__global__ void fill_kernel(float* dst, long ld, int col0, int ncols, int rows, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ncols) return;
  const int r = i / ncols, c = i - r * ncols;
  dst[(long)r * ld + col0 + c] = v;
}

int launch_fill(cpp_ctx* ctx, float* dst, long ld, int col0, int ncols, int rows, float v) {
  const int n = rows * ncols;
  if (n <= 0) return 0;
  prof_begin(ctx);
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dst, ld, col0,
                     ncols, rows, v);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// low-dim states: (rows, elems) f16|f32 -> f32 activation buffer with row stride ldd
__global__ void state_to_f32_kernel(float* dst, long ldd, const void* src, int dtype, long elems,
                                    int rows) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems * rows) return;
  const long r = i / elems, c = i - r * elems;
  dst[r * ldd + c] = dtype == 1 ? __half2float(((const __half*)src)[i]) : ((const float*)src)[i];
}

int launch_state_to_f32(cpp_ctx* ctx, float* dst, long ldd, const void* src, int dtype, long elems,
                        int rows) {
  const long n = elems * rows;
  prof_begin(ctx);
  hipLaunchKernelGGL(state_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     ctx->stream, dst, ldd, src, dtype, elems, rows);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// grad_ys of ddpg_cartpole.py:111-113 pushed through the tanh head: dz = -dq_da * (1 - a^2)
__global__ void actor_head_grad_kernel(float* dz, const float* dq_da, const float* act, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = act[i];
  dz[i] = -dq_da[i] * (1.f - a * a);
}

int launch_actor_head_grad(cpp_ctx* ctx, float* dz, const float* dq_da, const float* act, int n) {
  prof_begin(ctx);
  hipLaunchKernelGGL(actor_head_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, dz,
                     dq_da, act, n);
  LAUNCH_CHECK();
  prof_end(ctx, K_ELEMENTWISE);
  return 0;
}

// ddpg_cartpole.py:199-209: y = r + (mask*discount)*Q'(s2, mu'(s2)); td = Q - y; loss = mean(td^2);
// dq = d loss / d Q = 2 td / B.
__global__ __launch_bounds__(256) void td_kernel(const float* q, const float* tq, const float* r,
                                                 const float* mask, float discount, int B, float* td,
                                                 float* dq, float* loss) {
  __shared__ double red[256];
  double s = 0.0;
  const float inv_b = 2.f / (float)B;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float y = r[i] + (mask[i] * discount) * tq[i];
    const float t = q[i] - y;
    td[i] = t;
    if (dq) dq[i] = t * inv_b;
    s += (double)t * (double)t;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)B);
}

int launch_td(cpp_ctx* ctx, const float* q, const float* tq, const float* r, const float* mask,
              float discount, int B, float* td, float* dq, float* loss) {
  prof_begin(ctx);
  hipLaunchKernelGGL(td_kernel, dim3(1), dim3(256), 0, ctx->stream, q, tq, r, mask, discount, B, td,
                     dq, loss);
  LAUNCH_CHECK();
  prof_end(ctx, K_TD);
  return 0;
}


// NAF head (naf_cartpole.py:186-230) forward + backward, one row per thread iteration:
//   L = [lower | exp(diag) | 0] from l_values; d = u - mu; z = L^T d; A = -1/2 |z|^2; Q = V + A;
//   y = r + (mask*discount)*V'(s2); td = Q - y; loss = mean(td^2)
// backward of the loss: dQ = 2 td / B; dz = -z dQ; dL[i][j] = d_i dz_j; dd = L dz;
//   dl(off-diag) = dL, dl(diag) = dL * exp(l); d_mu = -dd, pushed through tanh: * (1 - mu^2).
#define NAF_MAX_A 8
__global__ __launch_bounds__(256) void naf_head_kernel(const NafHeadArgs a) {
  __shared__ double red[256];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const int A = a.A, NL = A * (A + 1) / 2;
  const float inv_b = 2.f / (float)a.B;
  double s = 0.0;
  int mybad = 0;
  for (int b = threadIdx.x; b < a.B; b += 256) {
    float L[NAF_MAX_A][NAF_MAX_A], d[NAF_MAX_A], z[NAF_MAX_A];
    const float* lv = a.lv + (long)b * NL;
    for (int i = 0; i < A; ++i) {
      const int off = i * (i + 1) / 2;
      for (int j = 0; j < A; ++j) L[i][j] = 0.f;
      for (int j = 0; j < i; ++j) { L[i][j] = lv[off + j]; if (!isfinite(L[i][j])) mybad = 1; }
      L[i][i] = expf(lv[off + i]);
      if (!isfinite(lv[off + i]) || !isfinite(L[i][i])) mybad = 1;
      d[i] = a.action[(long)b * A + i] - a.mu[(long)b * A + i];
    }
    float zz = 0.f;
    for (int j = 0; j < A; ++j) {
      float t = 0.f;
      for (int i = j; i < A; ++i) t += L[i][j] * d[i];
      z[j] = t; zz += t * t;
    }
    const float adv = -0.5f * zz;
    const float q = a.value[b] + adv;
    const float y = a.reward[b] + (a.mask[b] * a.discount) * a.target_value[b];
    const float td = q - y;
    if (a.adv) a.adv[b] = adv;
    if (a.q) a.q[b] = q;
    if (a.td) a.td[b] = td;
    s += (double)td * (double)td;
    if (a.d_value) {
      const float dq = td * inv_b;
      a.d_value[b] = dq;
      float dz[NAF_MAX_A];
      for (int j = 0; j < A; ++j) dz[j] = -z[j] * dq;
      for (int i = 0; i < A; ++i) {
        const int off = i * (i + 1) / 2;
        float dd = 0.f;
        for (int j = 0; j <= i; ++j) dd += L[i][j] * dz[j];
        for (int j = 0; j < i; ++j) a.d_l[(long)b * NL + off + j] = d[i] * dz[j];
        a.d_l[(long)b * NL + off + i] = d[i] * dz[i] * L[i][i];
        const float m = a.mu[(long)b * A + i];
        a.d_mu_z[(long)b * A + i] = -dd * (1.f - m * m);
      }
    }
  }
  if (mybad) atomicOr(&bad, 1);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float loss = (float)(red[0] / (double)a.B);
    a.loss[0] = loss;
    if (a.nonfinite && (bad || !isfinite(loss))) a.nonfinite[0] = 1;
  }
}

int launch_naf_head(cpp_ctx* ctx, const NafHeadArgs& a) {
  if (a.A > NAF_MAX_A) { cpp_set_error("naf head: action_dim %d > %d", a.A, NAF_MAX_A); return 1; }
  prof_begin(ctx);
  hipLaunchKernelGGL(naf_head_kernel, dim3(1), dim3(256), 0, ctx->stream, a);
  LAUNCH_CHECK();
  prof_end(ctx, K_NAF_HEAD);
  return 0;
}


// ---- NAF heads, fused (NafHeadsArgs, common.h).  One lane per batch row, 256 rows per workgroup (one workgroup for the
// reference's batch sizes: no cross-workgroup step for the loss).  The rows of the representation (live and target) are staged
// through LDS (coalesced 16-byte loads, every global load of the kernel issued before the first one is used), the four weight
// matrices sit in LDS; the head arithmetic is naf_head_kernel's with action_dim a template value (registers, no scratch);
// d(representation) leaves the kernel as the lane's own 16-byte pieces.
// Measured (cfg4, rocprofv3): 14.5 us per launch against 5.2 (naf_head_kernel) + a forward GEMM level + three backward levels
// of ~6.7 us each.  In-kernel clock: 1.4 us to issue the loads, 2.5 until the rows are staged, 2.0 forward, 0.8 head, 2.5 backward:
// the two dot-product phases are bound by the CU's one LDS pipe (a 16-byte read costs 8 cycles even when every lane reads the same
// address).  A variant with the rows in registers (per-lane 16-byte loads, no LDS) and the weights through the scalar cache was
// slower (16 us: 64 cache lines per load instruction, a scalar round trip per chunk of weights).
// loss = mean(td^2): with several workgroups each writes its partial through and the last one to arrive adds them in order.
constexpr int NAFH_ROWS = 256, NAFH_THREADS = 256, NAFH_XI = 16, NAFH_WL = 4;
typedef float nafh_f4 __attribute__((ext_vector_type(4)));
typedef unsigned nafh_u4 __attribute__((ext_vector_type(4)));
template <int AT>
__global__ __launch_bounds__(NAFH_THREADS) void naf_heads_kernel(const NafHeadsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float nl[];
  constexpr int A = AT, NL = A * (A + 1) / 2, NO = 1 + A + NL, NO4 = (NO + 3) / 4;
  static_assert(NO + 1 <= 16, "head values per row");
  constexpr int OOB = 0x7FFFFF00;
  const int K = a.rep + 1, KP = (K + 3) & ~3, Q = KP >> 2;      // Q <= 16 (naf_heads_supported)
  float* xs = nl;                       float* xts = xs + NAFH_ROWS * KP;      // [256][KP]
  float* wT = xts + NAFH_ROWS * KP;     // [16][KP]: row o = output o's weights over k (o = NO: the target value's), zero from K on
  float* wB = wT + 16 * KP;             // [KP][16]: unit j's weights towards the NO outputs (rows rep .. KP - 1 and columns NO .. 15: zero)
  __shared__ double lred[NAFH_THREADS / 64];
  __shared__ int lbad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row0 = blockIdx.x * NAFH_ROWS, b = row0 + tid;
  const bool rv = b < a.B;
  if (tid == 0) lbad = 0;
  if (tid == 0 && blockIdx.x == 0 && a.step_bump) *a.step_bump += 1ull;
  // ---- every global load, up front
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((long)a.B * a.ldx * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xt), 0, (int)((long)a.B * a.ldx * 4), 0x00020000);
  const int rpi = 64 / Q, rsub = lane / Q, q = lane - rsub * Q;          // rows per wave instruction; this lane's row in it, its 16-byte chunk
  const bool lv_ = rsub < rpi;
  nafh_u4 xv[NAFH_XI], xtv[NAFH_XI];
#pragma unroll
  for (int it = 0; it < NAFH_XI; ++it) {
    const int rr = wave * 64 + it * rpi + rsub;                          // (it * rpi + rsub < 64 for the lanes that count: rpi >= 4)
    const bool in = lv_ && it * rpi + rsub < 64;
    const int off = in ? (int)(((long)(row0 + rr) * a.ldx + 4 * q) * 4) : OOB;
    xv[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    xtv[it] = __builtin_amdgcn_raw_buffer_load_b128(rxt, off, 0, 0);
  }
  const __amdgpu_buffer_rsrc_t rwv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wv), 0, K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wvt), 0, K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwm = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wmu), 0, K * A * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wl), 0, K * NL * 4, 0x00020000);
  const float wv_r = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwv, tid * 4, 0, 0));       // K <= 64 < 256
  const float wvt_r = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, tid * 4, 0, 0));
  float wm_r[NAFH_WL], wl_r[NAFH_WL];
#pragma unroll
  for (int n = 0; n < NAFH_WL; ++n) {
    wm_r[n] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwm, (tid + n * NAFH_THREADS) * 4, 0, 0));
    wl_r[n] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwl, (tid + n * NAFH_THREADS) * 4, 0, 0));
  }
  float act[A];
#pragma unroll
  for (int i = 0; i < A; ++i) act[i] = rv ? a.action[(long)b * A + i] : 0.f;
  const float rew = rv ? a.reward[b] : 0.f, msk = rv ? a.mask[b] : 0.f;
  // ---- into LDS
  for (int i = tid; i < 32 * KP; i += NAFH_THREADS) wT[i] = 0.f;         // wT and wB (wT: rows NO + 1 .. 15 and k >= K stay zero)
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NAFH_XI; ++it) {
    const int rl = it * rpi + rsub;
    if (lv_ && rl < 64) {
      nafh_u4 v = xv[it], vt = xtv[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) if (4 * q + e >= K) { v[e] = 0u; vt[e] = 0u; }       // (the chunk's tail belongs to the next row)
      *reinterpret_cast<nafh_u4*>(xs + (wave * 64 + rl) * KP + 4 * q) = v;
      *reinterpret_cast<nafh_u4*>(xts + (wave * 64 + rl) * KP + 4 * q) = vt;
    }
  }
  if (tid < K) { wT[tid] = wv_r; wT[NO * KP + tid] = wvt_r; if (tid < a.rep) wB[tid * 16] = wv_r; }
#pragma unroll
  for (int n = 0; n < NAFH_WL; ++n) {
    const int e = tid + n * NAFH_THREADS;
    if (e < K * A) { const int k = e / A, i = e - k * A; wT[(1 + i) * KP + k] = wm_r[n]; if (k < a.rep) wB[k * 16 + 1 + i] = wm_r[n]; }
    if (e < K * NL) { const int k = e / NL, j = e - k * NL; wT[(1 + A + j) * KP + k] = wl_r[n]; if (k < a.rep) wB[k * 16 + 1 + A + j] = wl_r[n]; }
  }
  __syncthreads();
  // ---- forward heads of this lane's row: NO outputs from the live row, the target value from the target's row
  float hv[NO + 1];
  {
    float acc[NO + 1];
#pragma unroll
    for (int o = 0; o <= NO; ++o) acc[o] = 0.f;
    const float* xr = xs + tid * KP; const float* xtr = xts + tid * KP;
#pragma unroll 2
    for (int k = 0; k < KP; k += 4) {
      const nafh_f4 x4 = *reinterpret_cast<const nafh_f4*>(xr + k), t4 = *reinterpret_cast<const nafh_f4*>(xtr + k);
#pragma unroll
      for (int o = 0; o <= NO; ++o) {
        const nafh_f4 w4 = *reinterpret_cast<const nafh_f4*>(wT + o * KP + k);       // (one address for the whole wave)
        const nafh_f4 u4 = o == NO ? t4 : x4;
        acc[o] = fmaf(u4[0], w4[0], acc[o]); acc[o] = fmaf(u4[1], w4[1], acc[o]);
        acc[o] = fmaf(u4[2], w4[2], acc[o]); acc[o] = fmaf(u4[3], w4[3], acc[o]);
      }
    }
#pragma unroll
    for (int o = 0; o <= NO; ++o) hv[o] = acc[o];
  }
  // ---- naf_head_kernel's row (naf_cartpole.py:186-230)
  double s2 = 0.0;
  float dzr[4 * NO4];
#pragma unroll
  for (int i = 0; i < 4 * NO4; ++i) dzr[i] = 0.f;
  if (rv) {
    const float value = hv[0], tvalue = hv[NO];
    float mu[A], L[A][A], d[A], z[A];
    int mybad = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) mu[i] = tanhf(hv[1 + i]);
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const int off = i * (i + 1) / 2;
#pragma unroll
      for (int j = 0; j < A; ++j) L[i][j] = 0.f;
#pragma unroll
      for (int j = 0; j < i; ++j) { L[i][j] = hv[1 + A + off + j]; if (!isfinite(L[i][j])) mybad = 1; }
      L[i][i] = expf(hv[1 + A + off + i]);
      if (!isfinite(hv[1 + A + off + i]) || !isfinite(L[i][i])) mybad = 1;
      d[i] = act[i] - mu[i];
    }
    float zz = 0.f;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float t = 0.f;
#pragma unroll
      for (int i = j; i < A; ++i) t += L[i][j] * d[i];
      z[j] = t; zz += t * t;
    }
    const float adv = -0.5f * zz, qv = value + adv;
    const float y = rew + (msk * a.discount) * tvalue;
    const float td = qv - y;
    a.value[b] = value; a.target_value[b] = tvalue;
#pragma unroll
    for (int i = 0; i < A; ++i) a.mu[(long)b * A + i] = mu[i];
#pragma unroll
    for (int j = 0; j < NL; ++j) a.lv[(long)b * NL + j] = hv[1 + A + j];
    if (a.adv) a.adv[b] = adv;
    if (a.q) a.q[b] = qv;
    if (a.td) a.td[b] = td;
    s2 = (double)td * (double)td;
    const float dq = td * (2.f / (float)a.B);
    a.d_value[b] = dq;
    float dz[A];
#pragma unroll
    for (int j = 0; j < A; ++j) dz[j] = -z[j] * dq;
    dzr[0] = dq;
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const int off = i * (i + 1) / 2;
      float dd = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) dd += L[i][j] * dz[j];
#pragma unroll
      for (int j = 0; j < i; ++j) dzr[1 + A + off + j] = d[i] * dz[j];
      dzr[1 + A + off + i] = d[i] * dz[i] * L[i][i];
      dzr[1 + i] = -dd * (1.f - mu[i] * mu[i]);
    }
#pragma unroll
    for (int i = 0; i < A; ++i) a.d_mu_z[(long)b * A + i] = dzr[1 + i];
#pragma unroll
    for (int j = 0; j < NL; ++j) a.d_l[(long)b * NL + j] = dzr[1 + A + j];
    if (mybad) atomicOr(&lbad, 1);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
  if (lane == 0) lred[wave] = s2;
  // ---- d(representation) of this lane's row: contributions in the order value, mu, l_values; ReLU mask from the live row; out as
  // the lane's own 16-byte pieces (rows are 4-byte aligned: buffer stores take that; the last piece of a row element by element).
  // Four chunks of units per round, every LDS read of a round in front of its arithmetic (one wave per SIMD: nothing else hides
  // the LDS latency); the last round repeats the last chunk.
  {
    const float* xr = xs + tid * KP;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(a.drep, 0, (int)((long)a.B * a.ldd * 4), 0x00020000);
    const int dbase = (int)((long)b * a.ldd * 4);
    const float two = a.epi == GE_MUL_RELU_GRAD_X2 ? 2.f : 1.f;
    for (int c0 = 0; c0 < Q; c0 += 4) {
      nafh_f4 x4[4], w4[4][4][NO4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = c0 + cc < Q ? c0 + cc : Q - 1;
        x4[cc] = *reinterpret_cast<const nafh_f4*>(xr + 4 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < NO4; ++i) w4[cc][e][i] = *reinterpret_cast<const nafh_f4*>(wB + (4 * c + e) * 16 + 4 * i);   // (rows rep .. KP - 1: zero)
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = c0 + cc < Q ? c0 + cc : Q - 1;
        nafh_u4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float acc = dzr[0] * w4[cc][e][0][0];
#pragma unroll
          for (int o = 1; o < NO; ++o) acc = fmaf(dzr[o], w4[cc][e][o >> 2][o & 3], acc);
          o4[e] = __float_as_uint(x4[cc][e] > 0.f ? two * acc : 0.f);
        }
        if (4 * c + 3 < a.rep) __builtin_amdgcn_raw_buffer_store_b128(o4, rd, rv ? dbase + 16 * c : OOB, 0, 0);
        else {
#pragma unroll
          for (int e = 0; e < 3; ++e)
            if (4 * c + e < a.rep) __builtin_amdgcn_raw_buffer_store_b32(o4[e], rd, rv ? dbase + 16 * c + 4 * e : OOB, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  // ---- loss: this workgroup's partial; with several workgroups it is written through and the last one to arrive adds them in order
  if (tid == 0) {
    double p = 0.0;
    for (int i = 0; i < NAFH_THREADS / 64; ++i) p += lred[i];
    double sum = p; int bad = lbad; bool last = gridDim.x == 1;
    if (!last) {
      __hip_atomic_store(a.part + blockIdx.x, p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.part + NAF_HEADS_MAX_WGS + blockIdx.x, lbad ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1u == gridDim.x) {
        last = true; sum = 0.0; bad = 0;
        for (unsigned i = 0; i < gridDim.x; ++i) {
          sum += __hip_atomic_load(a.part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad |= __hip_atomic_load(a.part + NAF_HEADS_MAX_WGS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0;
        }
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (last) {
      const float loss = (float)(sum / (double)a.B);
      a.loss[0] = loss;
      if (a.nonfinite && (bad || !isfinite(loss))) a.nonfinite[0] = 1;
    }
  }
}

static size_t naf_heads_lds(const NafHeadsArgs& a) {
  const size_t K = a.rep + 1, KP = (K + 3) & ~(size_t)3;
  return (2 * NAFH_ROWS * KP + 16 * KP + KP * 16) * sizeof(float);
}

bool naf_heads_supported(const NafHeadsArgs& a) {
  const int NLx = a.A * (a.A + 1) / 2, K = a.rep + 1;
  return a.A >= 1 && a.A <= 4 && a.rep >= 1 && K <= 64 && K * NLx <= NAFH_WL * NAFH_THREADS && a.drep && a.Y == a.x && a.ldy == a.ldx &&
         (a.B + NAFH_ROWS - 1) / NAFH_ROWS <= NAF_HEADS_MAX_WGS && naf_heads_lds(a) <= 150 * 1024;
}

int launch_naf_heads(cpp_ctx* ctx, const NafHeadsArgs& a) {
  const size_t lds = naf_heads_lds(a);
  typedef void (*kern_t)(const NafHeadsArgs);
  static const kern_t kerns[4] = {naf_heads_kernel<1>, naf_heads_kernel<2>, naf_heads_kernel<3>, naf_heads_kernel<4>};
  if (a.A < 1 || a.A > 4) { cpp_set_error("naf heads: action_dim %d", a.A); return 1; }
  static size_t attr[CPP_MAX_DEVICES][4] = {};
  size_t& have = attr[cpp_dev_slot(ctx)][a.A - 1];
  if (lds > have) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kerns[a.A - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
  }
  prof_begin(ctx);
  hipLaunchKernelGGL(kerns[a.A - 1], dim3((a.B + NAFH_ROWS - 1) / NAFH_ROWS), dim3(NAFH_THREADS), lds, ctx->stream, a);
  LAUNCH_CHECK();
  prof_end(ctx, K_NAF_HEAD);
  return 0;
}

// ---- NAF, shared representation with TWO hidden layers (the reference's pixel NAF: 100, 50): everything between the first hidden
// layer's activations and the one backward GEMM level that is left, row-local, on the matrix pipes (NafMlpArgs, common.h).
// A workgroup owns 16 batch rows (grid = ceil(B / 16)); its four waves each own one 16-column tile:
//   1. h1 = relu([h0, 1] [W1; b1]) for the live and the target network (26 MFMA steps each, operands straight from global memory);
//   2. the four head layers from h1 (through LDS: accumulator layout -> A-operand layout), 13 steps (wave 0: value, mu, l_values;
//      wave 1: the target's value);
//   3. naf_head_kernel's arithmetic, one lane per row (wave 0);
//   4. dz1 = relu'(h1) (dz_heads [Wv | Wmu | Wl]^T) -- the contributions in the MFMA's k order value, mu.., l.. -- and
//   5. dz0 = relu'(h0) (dz1 W1^T), 13 steps per tile, two tiles per wave.
// Every global load is issued before the first use.  loss = mean(td^2): a partial per workgroup, written through; the last workgroup
// to arrive adds them in order (at most B / 16 of them).
constexpr int NAFM_S1 = 26, NAFM_S2 = 13, NAFM_LD = 52, NAFM_XS = 105, NAFM_XL = 7;   // XL: ceil(16 * 104 / 256) staged elements per thread
// S1, S2:      // k steps of layer 1 (n0 + 1 <= 104) and of the heads / dX (n1 + 1 <= 52); LDS row stride
template <int AT>
__global__ __launch_bounds__(256) void naf_mlp_kernel(const NafMlpArgs m) {
  const NafHeadsArgs& a = m.h;
  constexpr int A = AT, NL = A * (A + 1) / 2, NO = 1 + A + NL;
  static_assert(NO <= 15, "head values per row");
  constexpr int OOB = 0x7FFFFF00;
  __shared__ __attribute__((aligned(16))) float h1s[16][NAFM_LD], h1ts[16][NAFM_LD], dz1s[16][NAFM_LD], outs[16][16], dzs[16][16], outt[16];
  __shared__ float x0s[16][NAFM_XS], x0ts[16][NAFM_XS];      // h0 rows; odd stride: the A-operand reads of 16 rows fall on different banks
  __shared__ int lbad;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lj = lane >> 4, r0 = blockIdx.x * 16;
  const int n0 = m.n0, n1 = a.rep, K1 = n0 + 1, K2 = n1 + 1;
  if (tid == 0) lbad = 0;
  if (tid == 0 && blockIdx.x == 0 && a.step_bump) *a.step_bump += 1ull;
  // ---- every global load, up front
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.x0), 0, (int)((long)a.B * m.ld0 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.x0t), 0, (int)((long)a.B * m.ld0 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.W1), 0, K1 * n1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(m.W1t), 0, K1 * n1 * 4, 0x00020000);
  // the 16 rows of h0 (live, target) go through LDS once per workgroup (each wave needs all of them as its A operand: read straight
  // from global memory by every wave, the 26 + 26 four-byte loads per lane touched 16 cache lines each -- 4.7 us of the CU's one
  // address path)
  float xg[NAFM_XL], xgt[NAFM_XL];
#pragma unroll
  for (int u = 0; u < NAFM_XL; ++u) {
    const int e = tid + 256 * u, rr = e / K1, k = e - rr * K1;
    const int ao = rr < 16 ? ((r0 + rr) * m.ld0 + k) * 4 : OOB;                    // (rows past the batch are behind the descriptor's end)
    xg[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, ao, 0, 0));
    xgt[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rxt, ao, 0, 0));
  }
  float b1[NAFM_S1], b1t[NAFM_S1];
  const int ncol = 16 * w + li;                              // this lane's column of layer 1 / unit of the representation
#pragma unroll
  for (int s = 0; s < NAFM_S1; ++s) {
    const int k = 4 * s + lj;
    const int bo = (k < K1 && ncol < n1) ? (k * n1 + ncol) * 4 : OOB;
    b1[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, bo, 0, 0));
    b1t[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rwt, bo, 0, 0));
  }
  // head weights as the B operand [k][o = li]: wave 0 the live heads, wave 1 the target's value (column 0 only).  A lane's column lives in
  // ONE of the four matrices; every lane asks all of them through bounded descriptors, with an offset behind the end where the matrix is
  // not its own (zero comes back), and adds what arrives: no branch around a load (as `cond ? p[i] : q[j]` each of the 13 loads sat in a
  // basic block of its own behind s_waitcnt vmcnt(0): 13 round trips, most of this kernel's first version).
  const __amdgpu_buffer_rsrc_t rv_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wv), 0, K2 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rvt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wvt), 0, K2 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rmu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wmu), 0, K2 * A * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Wl), 0, K2 * NL * 4, 0x00020000);
  const bool is_v = w == 0 && li == 0, is_vt = w == 1 && li == 0, is_mu = w == 0 && li >= 1 && li < 1 + A, is_l = w == 0 && li >= 1 + A && li < NO;
  float bh[NAFM_S2];
#pragma unroll
  for (int s = 0; s < NAFM_S2; ++s) {
    const int k = 4 * s + lj;
    const bool kin = k < K2;
    const float v0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv_, (is_v && kin) ? k * 4 : OOB, 0, 0));
    const float v1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rvt, (is_vt && kin) ? k * 4 : OOB, 0, 0));
    const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmu, (is_mu && kin) ? (k * A + li - 1) * 4 : OOB, 0, 0));
    const float v3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, (is_l && kin) ? (k * NL + li - 1 - A) * 4 : OOB, 0, 0));
    bh[s] = (v0 + v1) + (v2 + v3);                          // (three of the four are zero)
  }
  // dz1's B operand [k = o][n = unit j]: the heads' weights of unit j = ncol, the same way
  float bd[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int o = 4 * s + lj;
    const bool jin = ncol < n1;
    const float v0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rv_, (jin && o == 0) ? ncol * 4 : OOB, 0, 0));
    const float v2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rmu, (jin && o >= 1 && o < 1 + A) ? (ncol * A + o - 1) * 4 : OOB, 0, 0));
    const float v3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, (jin && o >= 1 + A && o < NO) ? (ncol * NL + o - 1 - A) * 4 : OOB, 0, 0));
    bd[s] = v0 + (v2 + v3);
  }
  // dz0's B operand [k = unit j][n = input unit mm] = W1[mm][j], two column tiles per wave; and h0 in the accumulator layout (the mask)
  float bx[2][NAFM_S2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int mm = 16 * (w + 4 * t2) + li;
#pragma unroll
    for (int s = 0; s < NAFM_S2; ++s) {
      const int j = 4 * s + lj;
      bx[t2][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, (mm < n0 && j < n1) ? (mm * n1 + j) * 4 : OOB, 0, 0));
    }
  }
  const int hrow = r0 + (lane & 15);
  const bool hrv = w == 0 && lane < 16 && hrow < a.B;        // the lanes that run the head arithmetic
  float act[A];
#pragma unroll
  for (int i = 0; i < A; ++i) act[i] = hrv ? a.action[(long)hrow * A + i] : 0.f;
  const float rew = hrv ? a.reward[hrow] : 0.f, msk = hrv ? a.mask[hrow] : 0.f;
  // LDS: zero, the bias inputs
  for (int i = tid; i < 16 * NAFM_LD; i += 256) {
    const float one = (i % NAFM_LD) == n1 ? 1.f : 0.f;
    (&h1s[0][0])[i] = one; (&h1ts[0][0])[i] = one; (&dz1s[0][0])[i] = 0.f;
  }
  (&dzs[0][0])[tid] = 0.f; (&outs[0][0])[tid] = 0.f;
  if (tid < 16) outt[tid] = 0.f;
#pragma unroll
  for (int u = 0; u < NAFM_XL; ++u) {
    const int e = tid + 256 * u, rr = e / K1, k = e - rr * K1;
    if (rr < 16) { x0s[rr][k] = xg[u]; x0ts[rr][k] = xgt[u]; }
  }
  if (tid < 16) for (int k = K1; k < 4 * NAFM_S1; ++k) { x0s[tid][k] = 0.f; x0ts[tid][k] = 0.f; }      // (k steps past the row: zero)
  __syncthreads();
  // ---- 1. layer 1, live and target
  f32x4 c1 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1t = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NAFM_S1; ++s) { c1 = MFMA16(x0s[li][4 * s + lj], b1[s], c1); c1t = MFMA16(x0ts[li][4 * s + lj], b1t[s], c1t); }
  float h1v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * lj + i;
    h1v[i] = fmaxf(c1[i], 0.f);
    if (ncol < n1) {
      h1s[row][ncol] = h1v[i]; h1ts[row][ncol] = fmaxf(c1t[i], 0.f);
      if (r0 + row < a.B) m.h1_out[(long)(r0 + row) * m.ld1 + ncol] = h1v[i];
    }
  }
  __syncthreads();
  // ---- 2. the head layers
  if (w < 2) {
    const float (*src)[NAFM_LD] = w == 0 ? h1s : h1ts;
    f32x4 ch = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NAFM_S2; ++s) ch = MFMA16(src[li][4 * s + lj], bh[s], ch);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (w == 0) outs[4 * lj + i][li] = ch[i];
      else if (li == 0) outt[4 * lj + i] = ch[i];
    }
  }
  __syncthreads();
  // ---- 3. naf_head_kernel's row (naf_cartpole.py:186-230), one lane per row
  double s2 = 0.0;
  if (hrv) {
    const int b = hrow, l = lane;
    float hv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const f32x4 v4 = *reinterpret_cast<const f32x4*>(&outs[l][4 * i]); hv[4 * i] = v4[0]; hv[4 * i + 1] = v4[1]; hv[4 * i + 2] = v4[2]; hv[4 * i + 3] = v4[3]; }
    const float value = hv[0], tvalue = outt[l];
    float mu[A], L[A][A], d[A], z[A];
    int mybad = 0;
#pragma unroll
    for (int i = 0; i < A; ++i) mu[i] = tanhf(hv[1 + i]);
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const int off = i * (i + 1) / 2;
#pragma unroll
      for (int j = 0; j < A; ++j) L[i][j] = 0.f;
#pragma unroll
      for (int j = 0; j < i; ++j) { L[i][j] = hv[1 + A + off + j]; if (!isfinite(L[i][j])) mybad = 1; }
      L[i][i] = expf(hv[1 + A + off + i]);
      if (!isfinite(hv[1 + A + off + i]) || !isfinite(L[i][i])) mybad = 1;
      d[i] = act[i] - mu[i];
    }
    float zz = 0.f;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      float t = 0.f;
#pragma unroll
      for (int i = j; i < A; ++i) t += L[i][j] * d[i];
      z[j] = t; zz += t * t;
    }
    const float adv = -0.5f * zz, qv = value + adv;
    const float y = rew + (msk * a.discount) * tvalue;
    const float td = qv - y;
    a.value[b] = value; a.target_value[b] = tvalue;
#pragma unroll
    for (int i = 0; i < A; ++i) a.mu[(long)b * A + i] = mu[i];
#pragma unroll
    for (int j = 0; j < NL; ++j) a.lv[(long)b * NL + j] = hv[1 + A + j];
    if (a.adv) a.adv[b] = adv;
    if (a.q) a.q[b] = qv;
    if (a.td) a.td[b] = td;
    s2 = (double)td * (double)td;
    const float dq = td * (2.f / (float)a.B);
    a.d_value[b] = dq;
    float dz[A], dzr[NO];
#pragma unroll
    for (int j = 0; j < A; ++j) dz[j] = -z[j] * dq;
    dzr[0] = dq;
#pragma unroll
    for (int i = 0; i < A; ++i) {
      const int off = i * (i + 1) / 2;
      float dd = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) dd += L[i][j] * dz[j];
#pragma unroll
      for (int j = 0; j < i; ++j) dzr[1 + A + off + j] = d[i] * dz[j];
      dzr[1 + A + off + i] = d[i] * dz[i] * L[i][i];
      dzr[1 + i] = -dd * (1.f - mu[i] * mu[i]);
    }
#pragma unroll
    for (int i = 0; i < A; ++i) a.d_mu_z[(long)b * A + i] = dzr[1 + i];
#pragma unroll
    for (int j = 0; j < NL; ++j) a.d_l[(long)b * NL + j] = dzr[1 + A + j];
#pragma unroll
    for (int o = 0; o < NO; ++o) dzs[l][o] = dzr[o];
    if (mybad) atomicOr(&lbad, 1);
  }
  if (w == 0) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);       // (lanes 16 .. 63 hold zero)
  }
  __syncthreads();
  // ---- 4. dz1
  {
    f32x4 cd = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) cd = MFMA16(dzs[li][4 * s + lj], bd[s], cd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      const float v = h1v[i] > 0.f ? cd[i] : 0.f;
      if (ncol < n1) {
        dz1s[row][ncol] = v;
        if (r0 + row < a.B) a.drep[(long)(r0 + row) * a.ldd + ncol] = v;
      }
    }
  }
  __syncthreads();
  // ---- 5. dz0
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    const int mm = 16 * (w + 4 * t2) + li;
    if (16 * (w + 4 * t2) < n0) {                            // (uniform per wave)
      f32x4 cx = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NAFM_S2; ++s) cx = MFMA16(dz1s[li][4 * s + lj], bx[t2][s], cx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 4 * lj + i;
        if (mm < n0 && row < a.B) m.dz0[(long)row * n0 + mm] = x0s[4 * lj + i][mm] > 0.f ? cx[i] : 0.f;
      }
    }
  }
  // ---- loss: this workgroup's partial, written through; the last workgroup to arrive adds them in order
  if (tid == 0) {
    double sum = s2; int bad = lbad; bool last = gridDim.x == 1;
    if (!last) {
      __hip_atomic_store(a.part + blockIdx.x, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.part + NAF_HEADS_MAX_WGS + blockIdx.x, lbad ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t + 1u == gridDim.x) {
        last = true; sum = 0.0; bad = 0;
        for (unsigned i = 0; i < gridDim.x; ++i) {
          sum += __hip_atomic_load(a.part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bad |= __hip_atomic_load(a.part + NAF_HEADS_MAX_WGS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.0;
        }
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (last) {
      const float loss = (float)(sum / (double)a.B);
      a.loss[0] = loss;
      if (a.nonfinite && (bad || !isfinite(loss))) a.nonfinite[0] = 1;
    }
  }
}

bool naf_mlp_supported(const NafMlpArgs& m) {
  const NafHeadsArgs& a = m.h;
  return a.A >= 1 && a.A <= 4 && a.rep >= 1 && a.rep + 1 <= 4 * NAFM_S2 && a.rep <= 64 && m.n0 >= 1 && m.n0 + 1 <= 4 * NAFM_S1 && 16 * (m.n0 + 1) <= 256 * NAFM_XL && m.n0 <= 128 &&
         a.drep && m.dz0 && m.h1_out && a.ldd == a.rep && (a.B + 15) / 16 <= NAF_HEADS_MAX_WGS;
}

int launch_naf_mlp(cpp_ctx* ctx, const NafMlpArgs& m) {
  typedef void (*kern_t)(const NafMlpArgs);
  static const kern_t kerns[4] = {naf_mlp_kernel<1>, naf_mlp_kernel<2>, naf_mlp_kernel<3>, naf_mlp_kernel<4>};
  const int A = m.h.A;
  if (A < 1 || A > 4) { cpp_set_error("naf mlp: action_dim %d", A); return 1; }
  prof_begin(ctx);
  hipLaunchKernelGGL(kerns[A - 1], dim3((m.h.B + 15) / 16), dim3(256), 0, ctx->stream, m);
  LAUNCH_CHECK();
  prof_end(ctx, K_NAF_HEAD);
  return 0;
}

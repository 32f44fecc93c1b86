// DDPG "heads" kernel: everything between the last hidden layers of the four networks and the first backward GEMMs,
// row-local work that would otherwise be five dependent GEMM levels of a few thousand multiply-adds per row.
//
// Per batch row (ddpg_cartpole.py:95-100, :111-113, :166-171, :180-184, :199-209, :222):
//   a   = tanh(h2a Wo + bo)                               actor head            a' likewise from the target actor
//   h3  = relu([h2c, a] W3 + b3),  dQ/da at a = mu(s1)     critic, 2nd evaluation -> actor's grad_ys = -dQ/da through tanh
//   q   = relu([h2c, a_batch] W3 + b3) wq + bq             critic on the fed actions
//   q'  = relu([h2c', a'] W3' + b3') wq' + bq'              target critic at the target actor's action
//   td  = q - (r + mask discount q'),  loss = mean(td^2),  dz_q = 2 td / B
//   and one layer of both backward passes: dz of the critic's concat layer and of the layers feeding the two heads.
// A team of 16 lanes works on one row, a workgroup of 128 threads on 8 rows (grid = B / 8); the weights it needs sit
// in LDS (concat layer padded to a multiple of 16 columns so that a lane's 4 units need no guards).  The batch loss is
// left as one partial per workgroup; whoever reads the loss adds them in order.
#include "common.h"

constexpr int HEADS_THREADS = 128, HEADS_TEAM = 16, HEADS_ROWS = HEADS_THREADS / HEADS_TEAM, HEADS_AMAX = 8, HEADS_NU = 4;
constexpr int HEADS_NW4 = 10;                          // 16-byte chunks of [W3; b3] per thread: (n2c + A + 1) * n3 <= 40 * 128
constexpr int HEADS_N3P = HEADS_TEAM * HEADS_NU;       // lane t owns units 4t .. 4t+3 of the concat layer (n3 <= 64)
constexpr int HEADS_WSLACK = 64;                       // floats after [W3; b3] in LDS: units >= n3 read on into the next row
typedef float heads_f4 __attribute__((ext_vector_type(4)));
typedef float heads_f2 __attribute__((ext_vector_type(2)));
// [W3; b3] sits in LDS exactly as in memory (row stride n3): a flat 16-byte copy.  Rows are then only 8-byte aligned
// (n3 even), so a lane reads its 4 units as two ds_read_b64; units >= n3 pick up finite garbage that only ever meets
// zero weights of the q layer.
__device__ __forceinline__ heads_f4 ld4(const float* p) {
  const heads_f2 lo = *reinterpret_cast<const heads_f2*>(p), hi = *reinterpret_cast<const heads_f2*>(p + 2);
  return (heads_f4){lo[0], lo[1], hi[0], hi[1]};
}

// sum over the 16 lanes of a team, result in every lane: DPP row operations (a row IS 16 lanes), no LDS round trips
__device__ __forceinline__ float team_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));     // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));     // row_mirror
  return v;
}

__global__ __launch_bounds__(HEADS_THREADS) void ddpg_heads_kernel(const DdpgHeadsArgs h) {
  extern __shared__ __attribute__((aligned(16))) float hl[];
  constexpr int N3P = HEADS_N3P;
  const int A = h.A, n2a = h.n2a, n2c = h.n2c, n3 = h.n3;
  const int n2cp = (n2c + 3) & ~3;                     // x rows padded with zeros to float4s
  const int k3 = n2c + A + 1;                          // rows of [W3; b3]
  const int WS = n3;                                   // LDS row stride of [W3; b3] = the one in memory
  const int wfl = (k3 * n3 + HEADS_WSLACK + 3) & ~3;   // floats per weight image
  float* W3 = hl;                  float* W3t = W3 + wfl;
  float* wq = W3t + wfl;           float* wqt = wq + (N3P + 4);       // [0, N3P): weights (zero padded); N3P: the bias
  float* Wo = wqt + (N3P + 4);     float* Wot = Wo + (n2a + 1) * A;
  float* rowbase = hl + ((2 * wfl + 2 * (N3P + 4) + 2 * (n2a + 1) * A + 3) & ~3);
  const int rowf = N3P + 2 * n2cp + ((2 * n2a + 3) & ~3);  // per row: h3 / dz3 scratch, xc, xtc (16-byte aligned), xa, xta
  const int tid = threadIdx.x, t = tid & (HEADS_TEAM - 1), team = tid / HEADS_TEAM;
#ifdef HEADS_CLOCK
  unsigned long long ck[8]; int nck = 0;
#define HCK() ck[nck++] = __builtin_amdgcn_s_memrealtime()
#else
#define HCK()
#endif
  HCK();
  float* sc3 = rowbase + team * rowf; float* xc = sc3 + N3P; float* xtc = xc + n2cp; float* xa = xtc + n2cp; float* xta = xa + n2a;
  const int row = blockIdx.x * HEADS_ROWS + team;
  const bool rv = row < h.B;
  // every global load of the kernel is issued here, before the first use (a round trip to another XCD's L2 is ~2 us:
  // one batch of loads instead of a chain of them)
  constexpr int NX = HEADS_NU;                         // row inputs per lane (n2a, n2c <= 64)
  typedef unsigned heads_u4 __attribute__((ext_vector_type(4)));
  heads_u4 wv[HEADS_NW4], wtv[HEADS_NW4];
  float xav[NX], xtav[NX], xcv[NX], xtcv[NX], abv[HEADS_AMAX];
  {   // flat 16-byte chunks (4-byte aligned addresses are fine for buffer loads; reads past the end return 0)
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h.W3), 0, k3 * n3 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h.W3_t), 0, k3 * n3 * 4, 0x00020000);
#pragma unroll
    for (int n = 0; n < HEADS_NW4; ++n) {
      const int i = tid + n * HEADS_THREADS;
      wv[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, i * 16, 0, 0);
      wtv[n] = __builtin_amdgcn_raw_buffer_load_b128(rwt, i * 16, 0, 0);
    }
  }
#pragma unroll
  for (int n = 0; n < NX; ++n) {
    const int k = t + HEADS_TEAM * n;
    xav[n] = (rv && k < n2a) ? h.h2a[(long)row * h.ld_h2a + k] : 0.f;
    xtav[n] = (rv && k < n2a) ? h.h2ta[(long)row * h.ld_h2a + k] : 0.f;
    xcv[n] = (rv && k < n2c) ? h.h2c[(long)row * h.ld_h2c + k] : 0.f;
    xtcv[n] = (rv && k < n2c) ? h.h2tc[(long)row * h.ld_h2c + k] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < HEADS_AMAX; ++i) abv[i] = (rv && i < A) ? h.act[(long)row * A + i] : 0.f;
  const float rrow = rv ? h.r[row] : 0.f, mrow = rv ? h.mask[row] : 0.f;
  static_assert(N3P + 1 <= HEADS_THREADS, "one q-layer weight per thread");
  const float wq_a = tid < n3 ? h.wq[tid] : (tid == N3P ? h.wq[n3] : 0.f), wqt_a = tid < n3 ? h.wq_t[tid] : (tid == N3P ? h.wq_t[n3] : 0.f);
  float wov[4], wotv[4];                               // (n2a + 1) * A <= 4 * 128
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int i = tid + n * HEADS_THREADS;
    wov[n] = i < (n2a + 1) * A ? h.Wo[i] : 0.f; wotv[n] = i < (n2a + 1) * A ? h.Wo_t[i] : 0.f;
  }
  // the small operands first: the actor heads run while the concat-layer weights are still on their way
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    const int i = tid + n * HEADS_THREADS;
    if (i < (n2a + 1) * A) { Wo[i] = wov[n]; Wot[i] = wotv[n]; }
  }
#pragma unroll
  for (int n = 0; n < NX; ++n) {
    const int k = t + HEADS_TEAM * n;
    if (k < n2a) { xa[k] = xav[n]; xta[k] = xtav[n]; }
    if (k < n2cp) { xc[k] = xcv[n]; xtc[k] = xtcv[n]; }          // (zeros beyond n2c)
  }
  if (tid < N3P + 1) { wq[tid] = wq_a; wqt[tid] = wqt_a; }
  __syncthreads();
  HCK();
  float a[HEADS_AMAX], at[HEADS_AMAX], ab[HEADS_AMAX], dqda[HEADS_AMAX], adz[HEADS_AMAX];
  // ---- the two actor heads
#pragma unroll
  for (int i = 0; i < HEADS_AMAX; ++i) {
    a[i] = at[i] = ab[i] = dqda[i] = adz[i] = 0.f;
    if (i < A) {
      float s = 0.f, st = 0.f;
#pragma unroll 4
      for (int k = t; k < n2a; k += HEADS_TEAM) { s = fmaf(xa[k], Wo[k * A + i], s); st = fmaf(xta[k], Wot[k * A + i], st); }
      a[i] = tanhf(team_sum(s) + Wo[n2a * A + i]);
      at[i] = tanhf(team_sum(st) + Wot[n2a * A + i]);
      ab[i] = abv[i];
    }
  }
  HCK();
#pragma unroll
  for (int n = 0; n < HEADS_NW4; ++n) {
    const int i = tid + n * HEADS_THREADS;
    if (i * 4 < wfl) { reinterpret_cast<heads_u4*>(W3)[i] = wv[n]; reinterpret_cast<heads_u4*>(W3t)[i] = wtv[n]; }
  }
  __syncthreads();
  // ---- concat layer of the critic: three evaluations sharing the state part; lane t owns units 4t .. 4t+3; q, q'; dQ/da
  float qb = 0.f, qt = 0.f;
  {
    heads_f4 p = ld4(W3 + (n2c + A) * WS + 4 * t), pt = ld4(W3t + (n2c + A) * WS + 4 * t);
#pragma unroll 2
    for (int k = 0; k < n2cp; k += 4) {
      const heads_f4 xv = *reinterpret_cast<const heads_f4*>(xc + k), xtv = *reinterpret_cast<const heads_f4*>(xtc + k);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (k + kk < n2c) {
          const heads_f4 w = ld4(W3 + (k + kk) * WS + 4 * t), wt = ld4(W3t + (k + kk) * WS + 4 * t);
          p += xv[kk] * w; pt += xtv[kk] * wt;
        }
      }
    }
    heads_f4 pm = p, pb = p;
#pragma unroll
    for (int i = 0; i < HEADS_AMAX; ++i)
      if (i < A) {
        const heads_f4 w = ld4(W3 + (n2c + i) * WS + 4 * t), wt = ld4(W3t + (n2c + i) * WS + 4 * t);
        pm += a[i] * w; pb += ab[i] * w; pt += at[i] * wt;
      }
    const heads_f4 wqv = *reinterpret_cast<const heads_f4*>(wq + 4 * t), wqtv = *reinterpret_cast<const heads_f4*>(wqt + 4 * t);
    heads_f4 h3b, dzm;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      h3b[u] = fmaxf(pb[u], 0.f);
      qb = fmaf(h3b[u], wqv[u], qb); qt = fmaf(fmaxf(pt[u], 0.f), wqtv[u], qt);
      dzm[u] = pm[u] > 0.f ? wqv[u] : 0.f;             // dz of the concat layer on the 2nd evaluation (dz of q is 1)
      if (rv && 4 * t + u < n3) h.h3_out[(long)row * h.ld_h3 + 4 * t + u] = h3b[u];
    }
#pragma unroll
    for (int i = 0; i < HEADS_AMAX; ++i)
      if (i < A) {
        const heads_f4 w = ld4(W3 + (n2c + i) * WS + 4 * t);
        dqda[i] = (dzm[0] * w[0] + dzm[1] * w[1]) + (dzm[2] * w[2] + dzm[3] * w[3]);
      }
    HCK();
    qb = team_sum(qb) + wq[N3P]; qt = team_sum(qt) + wqt[N3P];
#pragma unroll
    for (int i = 0; i < HEADS_AMAX; ++i)
      if (i < A) { dqda[i] = team_sum(dqda[i]); adz[i] = -dqda[i] * (1.f - a[i] * a[i]); }
    const float td = rv ? qb - (rrow + (mrow * h.discount) * qt) : 0.f;
    const float dzq = td * (2.f / (float)h.B);
    if (rv && t == 0) { h.td[row] = td; h.dzq[row] = dzq; h.q_out[row] = qb; h.tq_out[row] = qt; }
    if (rv && t < A) {
#pragma unroll
      for (int i = 0; i < HEADS_AMAX; ++i)
        if (i == t) {
          h.a_out[(long)row * A + i] = a[i]; h.dq_da[(long)row * A + i] = dqda[i]; h.adz[(long)row * A + i] = adz[i];
          h.cat_splice[(long)row * h.ld_h2c + i] = ab[i];
        }
    }
    HCK();
    // ---- one layer back: the actor's last hidden layer, the critic's concat layer and the layer feeding it
    const float two = h.relu_x2 ? 2.f : 1.f;
    if (rv) {
      for (int k = t; k < n2a; k += HEADS_TEAM) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < HEADS_AMAX; ++i)
          if (i < A) s = fmaf(adz[i], Wo[k * A + i], s);
        h.dz_h2a[(long)row * n2a + k] = xa[k] > 0.f ? two * s : 0.f;
      }
    }
    heads_f4 d3;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d3[u] = h3b[u] > 0.f ? dzq * wqv[u] : 0.f;
      if (rv && 4 * t + u < n3) h.dz3[(long)row * n3 + 4 * t + u] = d3[u];
    }
    *reinterpret_cast<heads_f4*>(sc3 + 4 * t) = d3;
    HCK();
    __syncthreads();                                   // dz3 of the rows complete
    if (rv) {
      for (int k = t; k < n2c; k += HEADS_TEAM) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;    // padded columns hold zeros on both sides
#pragma unroll 4
        for (int j = 0; j < N3P; j += 4) {
          const heads_f4 dv = *reinterpret_cast<const heads_f4*>(sc3 + j), wv4 = ld4(W3 + k * WS + j);
          s0 = fmaf(dv[0], wv4[0], s0); s1 = fmaf(dv[1], wv4[1], s1); s2 = fmaf(dv[2], wv4[2], s2); s3 = fmaf(dv[3], wv4[3], s3);
        }
        h.dz2c[(long)row * n2c + k] = xc[k] > 0.f ? (s0 + s1) + (s2 + s3) : 0.f;
      }
    }
    HCK();
    // ---- loss = mean(td^2): one partial per workgroup; the reader adds the partials in order (cpp_ddpg_last_stats; no
    // second round trip through memory here)
    __shared__ double lred[HEADS_ROWS];
    if (t == 0) lred[team] = (double)td * (double)td;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int i = 0; i < HEADS_ROWS; ++i) s += lred[i];
      h.loss_part[blockIdx.x] = s;
    }
  }
#ifdef HEADS_CLOCK
  HCK();
  if (tid == 0 && blockIdx.x == 3) printf("HEADSCLK %llu %llu %llu %llu %llu %llu %llu (10 ns ticks)\n", ck[1]-ck[0], ck[2]-ck[1], ck[3]-ck[2], ck[4]-ck[3], ck[5]-ck[4], ck[6]-ck[5], ck[7]-ck[6]);
#endif
}

size_t ddpg_heads_lds_bytes(const DdpgHeadsArgs& h) {
  const size_t k3 = h.n2c + h.A + 1, n3p = HEADS_N3P, n2cp = (h.n2c + 3) & ~3;
  const size_t wfl = (k3 * h.n3 + HEADS_WSLACK + 3) & ~(size_t)3;
  const size_t w = (2 * wfl + 2 * (n3p + 4) + 2 * (size_t)(h.n2a + 1) * h.A + 3) & ~(size_t)3;
  const size_t f = w + (size_t)HEADS_ROWS * (n3p + 2 * n2cp + ((2 * h.n2a + 3) & ~3));
  return f * sizeof(float);
}

bool ddpg_heads_supported(const DdpgHeadsArgs& h) {
  return h.A <= HEADS_AMAX && h.n3 <= HEADS_TEAM * HEADS_NU && h.n2a <= HEADS_TEAM * HEADS_NU && h.n2c <= HEADS_TEAM * HEADS_NU &&
         (h.n2a + 1) * h.A <= 4 * HEADS_THREADS && (h.n2c + h.A + 1) * h.n3 + HEADS_WSLACK + 4 <= 4 * HEADS_NW4 * HEADS_THREADS && (h.n3 & 1) == 0 && ddpg_heads_lds_bytes(h) <= 120 * 1024 && (h.B + HEADS_ROWS - 1) / HEADS_ROWS <= DDPG_HEADS_MAX_WGS;
}

int launch_ddpg_heads(cpp_ctx* ctx, const DdpgHeadsArgs& h) {
  const size_t lds = ddpg_heads_lds_bytes(h);
  static size_t attr = 0;
  if (lds > attr) {
    HIP_CHECK(hipFuncSetAttribute((const void*)ddpg_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = lds;
  }
  prof_begin(ctx);
  hipLaunchKernelGGL(ddpg_heads_kernel, dim3((h.B + HEADS_ROWS - 1) / HEADS_ROWS), dim3(HEADS_THREADS), lds, ctx->stream, h);
  LAUNCH_CHECK();
  prof_end(ctx, K_TD);
  return 0;
}

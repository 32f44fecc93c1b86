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
// A team of 64 lanes (one wave) works on one row, a workgroup of 256 threads on 4 rows (grid = B / 4); the weights it needs sit
// in LDS (lane t owns unit t of the concat layer and element t of every row vector).  The batch loss is
// left as one partial per workgroup; whoever reads the loss adds them in order.
#include "common.h"

constexpr int HEADS_THREADS = 256, HEADS_TEAM = 64, HEADS_ROWS = HEADS_THREADS / HEADS_TEAM, HEADS_AMAX = 8;
constexpr int HEADS_NW4 = 5;                           // 16-byte chunks of [W3; b3] per thread: (n2c + A + 1) * n3 + slack <= 20 * 256
constexpr int HEADS_NW4P = 5;                          // the same for [W2; b2] of the optional actor layer
constexpr int HEADS_N1MAX = 2 * HEADS_TEAM;            // its inputs: two per lane
constexpr int HEADS_N3P = HEADS_TEAM;                  // lane t of a team owns unit t of the concat layer (n3 <= 64)
constexpr int HEADS_WSLACK = 64;                       // floats after [W3; b3] in LDS: units >= n3 read on into the next row
typedef float heads_f4 __attribute__((ext_vector_type(4)));
typedef float heads_f2 __attribute__((ext_vector_type(2)));
// [W3; b3] sits in LDS exactly as in memory (row stride n3): a flat 16-byte copy.  Rows are then only 8-byte aligned
// (n3 even); units >= n3 pick up finite garbage that only ever meets zero weights of the q layer.
__device__ __forceinline__ heads_f4 ld4(const float* p) {
  const heads_f2 lo = *reinterpret_cast<const heads_f2*>(p), hi = *reinterpret_cast<const heads_f2*>(p + 2);
  return (heads_f4){lo[0], lo[1], hi[0], hi[1]};
}

// sum over the 64 lanes of a team (= one wave), result in every lane: DPP row operations for the 16-lane rows, two
// cross-row exchanges; fixed order
__device__ __forceinline__ float team_sum(float v) {
  int x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));     // row_half_mirror
  x = __float_as_int(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));     // row_mirror
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// AT: compile-time bound of the action loops; EXACT: A == AT (the loops then carry no branches that keep the compiler from
// batching their LDS reads and interleaving the team sums)
template <int AT, bool EXACT>
__global__ __launch_bounds__(HEADS_THREADS) void ddpg_heads_kernel(const DdpgHeadsArgs h) {
  extern __shared__ __attribute__((aligned(16))) float hl[];
  constexpr int N3P = HEADS_N3P;
  const int A = EXACT ? AT : h.A, n2a = h.n2a, n2c = h.n2c, n3 = h.n3;
  const int n2cp = (n2c + 3) & ~3;                     // x rows padded with zeros to float4s
  const int k3 = n2c + A + 1;                          // rows of [W3; b3]
  const int WS = n3;                                   // LDS row stride of [W3; b3] = the one in memory
  const int wfl = (k3 * n3 + HEADS_WSLACK + 3) & ~3;   // floats per weight image
  float* W3 = hl;                  float* W3t = W3 + wfl;
  float* wq = W3t + wfl;           float* wqt = wq + (N3P + 4);       // [0, N3P): weights (zero padded); N3P: the bias
  float* Wo = wqt + (N3P + 4);     float* Wot = Wo + (n2a + 1) * A;
  const int n1a = h.n1a, n1ap = (n1a + 3) & ~3;
  const int w2fl = n1a ? (((n1a + 1) * n2a + HEADS_WSLACK + 3) & ~3) : 0;
  float* W2 = hl + ((2 * wfl + 2 * (N3P + 4) + 2 * (n2a + 1) * A + 3) & ~3); float* W2t = W2 + w2fl;
  float* rowbase = W2t + w2fl;
  const int rowf0 = N3P + 2 * n2cp + ((2 * n2a + 3) & ~3);  // per row: dz3 scratch, xc, xtc (16-byte aligned), xa, xta
  const int rowf = rowf0 + (n1a ? HEADS_TEAM + 2 * n1ap : 0);   // ... dz2 scratch, x1a, x1ta
  const int tid = threadIdx.x, t = tid & (HEADS_TEAM - 1), team = tid / HEADS_TEAM;
#ifdef HEADS_CLOCK
  unsigned long long ck[8]; int nck = 0;
#define HCK() ck[nck++] = __builtin_amdgcn_s_memrealtime()
#else
#define HCK()
#endif
  HCK();
  float* sc3 = rowbase + team * rowf; float* xc = sc3 + N3P; float* xtc = xc + n2cp;
  float* sc2 = sc3 + rowf0; float* x1 = sc2 + HEADS_TEAM; float* x1t = x1 + n1ap;
  const int row = blockIdx.x * HEADS_ROWS + team;
  const bool rv = row < h.B;
  // every global load of the kernel is issued here, before the first use (a round trip to another XCD's L2 is ~2 us:
  // one batch of loads instead of a chain of them)
  typedef unsigned heads_u4 __attribute__((ext_vector_type(4)));
  heads_u4 wv[HEADS_NW4], wtv[HEADS_NW4];
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
  heads_u4 w2v[HEADS_NW4P], w2tv[HEADS_NW4P];
  float x1v[2] = {0.f, 0.f}, x1tv[2] = {0.f, 0.f};
  float xav = 0.f, xtav = 0.f;                         // n2a, n2c <= 64: one element per lane
  if (n1a) {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h.W2), 0, (n1a + 1) * n2a * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(h.W2_t), 0, (n1a + 1) * n2a * 4, 0x00020000);
#pragma unroll
    for (int n = 0; n < HEADS_NW4P; ++n) {
      const int i = tid + n * HEADS_THREADS;
      w2v[n] = __builtin_amdgcn_raw_buffer_load_b128(rw, i * 16, 0, 0);
      w2tv[n] = __builtin_amdgcn_raw_buffer_load_b128(rwt, i * 16, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int j = t + n * HEADS_TEAM;
      x1v[n] = (rv && j < n1a) ? h.h1a[(long)row * h.ld_h1a + j] : 0.f;
      x1tv[n] = (rv && j < n1a) ? h.h1ta[(long)row * h.ld_h1a + j] : 0.f;
    }
  } else {
    xav = (rv && t < n2a) ? h.h2a[(long)row * h.ld_h2a + t] : 0.f;
    xtav = (rv && t < n2a) ? h.h2ta[(long)row * h.ld_h2a + t] : 0.f;
  }
  float xcv = 0.f, xtcv = 0.f;
  {
    xcv = (rv && t < n2c) ? h.h2c[(long)row * h.ld_h2c + t] : 0.f;
    xtcv = (rv && t < n2c) ? h.h2tc[(long)row * h.ld_h2c + t] : 0.f;
  }
  float abv[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) abv[i] = (rv && i < A) ? h.act[(long)row * A + i] : 0.f;
  const float rrow = rv ? h.r[row] : 0.f, mrow = rv ? h.mask[row] : 0.f;
  static_assert(N3P + 1 <= HEADS_THREADS, "one q-layer weight per thread");
  const float wq_a = tid < n3 ? h.wq[tid] : (tid == N3P ? h.wq[n3] : 0.f), wqt_a = tid < n3 ? h.wq_t[tid] : (tid == N3P ? h.wq_t[n3] : 0.f);
  float wov[2], wotv[2];                               // (n2a + 1) * A <= 2 * 256
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int i = tid + n * HEADS_THREADS;
    wov[n] = i < (n2a + 1) * A ? h.Wo[i] : 0.f; wotv[n] = i < (n2a + 1) * A ? h.Wo_t[i] : 0.f;
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int i = tid + n * HEADS_THREADS;
    if (i < (n2a + 1) * A) { Wo[i] = wov[n]; Wot[i] = wotv[n]; }
  }
  if (t < n2cp) { xc[t] = xcv; xtc[t] = xtcv; }        // (zeros beyond n2c)
  if (tid < N3P + 1) { wq[tid] = wq_a; wqt[tid] = wqt_a; }
#pragma unroll
  for (int n = 0; n < HEADS_NW4; ++n) {
    const int i = tid + n * HEADS_THREADS;
    if (i * 4 < wfl) { reinterpret_cast<heads_u4*>(W3)[i] = wv[n]; reinterpret_cast<heads_u4*>(W3t)[i] = wtv[n]; }
  }
  if (n1a) {
#pragma unroll
    for (int n = 0; n < HEADS_NW4P; ++n) {
      const int i = tid + n * HEADS_THREADS;
      if (i * 4 < w2fl) { reinterpret_cast<heads_u4*>(W2)[i] = w2v[n]; reinterpret_cast<heads_u4*>(W2t)[i] = w2tv[n]; }
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int j = t + n * HEADS_TEAM;
      if (j < n1ap) { x1[j] = x1v[n]; x1t[j] = x1tv[n]; }    // (zeros beyond n1a)
    }
  }
  __syncthreads();
  HCK();
  if (n1a) {   // ---- the actors' last hidden layer: lane t owns unit t (units >= n2a read on into finite weights and are zeroed)
    // (x is zero from n1a to n1ap and the rows it meets there -- the bias row, then the zero slack -- are finite: no bounds in the loop)
    float p2 = W2[n1a * n2a + t], p2t = W2t[n1a * n2a + t], q2 = 0.f, q2t = 0.f;
#pragma unroll 4
    for (int k = 0; k < n1ap; k += 4) {
      const heads_f4 xv = *reinterpret_cast<const heads_f4*>(x1 + k), xtv = *reinterpret_cast<const heads_f4*>(x1t + k);
      p2 = fmaf(xv[0], W2[(k + 0) * n2a + t], p2); p2t = fmaf(xtv[0], W2t[(k + 0) * n2a + t], p2t);
      q2 = fmaf(xv[1], W2[(k + 1) * n2a + t], q2); q2t = fmaf(xtv[1], W2t[(k + 1) * n2a + t], q2t);
      p2 = fmaf(xv[2], W2[(k + 2) * n2a + t], p2); p2t = fmaf(xtv[2], W2t[(k + 2) * n2a + t], p2t);
      q2 = fmaf(xv[3], W2[(k + 3) * n2a + t], q2); q2t = fmaf(xtv[3], W2t[(k + 3) * n2a + t], q2t);
    }
    xav = t < n2a ? fmaxf(p2 + q2, 0.f) : 0.f; xtav = t < n2a ? fmaxf(p2t + q2t, 0.f) : 0.f;
    if (rv && t < n2a) h.h2a_out[(long)row * h.ld_h2a + t] = xav;
  }
  float a[AT], at[AT], ab[AT], dqda[AT], adz[AT];
  // ---- the two actor heads: lane t holds x[t]
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    a[i] = at[i] = ab[i] = dqda[i] = adz[i] = 0.f;
    if (i < A) {
      const float s = t < n2a ? xav * Wo[t * A + i] : 0.f, st = t < n2a ? xtav * Wot[t * A + i] : 0.f;
      a[i] = tanhf(team_sum(s) + Wo[n2a * A + i]);
      at[i] = tanhf(team_sum(st) + Wot[n2a * A + i]);
      ab[i] = abv[i];
    }
  }
  HCK();
  // ---- concat layer of the critic: three evaluations sharing the state part; lane t owns unit t; q, q'; dQ/da
  float p = W3[(n2c + A) * WS + t], pt = W3t[(n2c + A) * WS + t], q = 0.f, qt_ = 0.f;
#pragma unroll 4
  for (int k = 0; k < n2cp; k += 4) {     // (x is zero from n2c to n2cp; the rows it meets there are finite)
    const heads_f4 xv = *reinterpret_cast<const heads_f4*>(xc + k), xtv = *reinterpret_cast<const heads_f4*>(xtc + k);
    p = fmaf(xv[0], W3[(k + 0) * WS + t], p); pt = fmaf(xtv[0], W3t[(k + 0) * WS + t], pt);
    q = fmaf(xv[1], W3[(k + 1) * WS + t], q); qt_ = fmaf(xtv[1], W3t[(k + 1) * WS + t], qt_);
    p = fmaf(xv[2], W3[(k + 2) * WS + t], p); pt = fmaf(xtv[2], W3t[(k + 2) * WS + t], pt);
    q = fmaf(xv[3], W3[(k + 3) * WS + t], q); qt_ = fmaf(xtv[3], W3t[(k + 3) * WS + t], qt_);
  }
  p += q; pt += qt_;
  float pm = p, pb = p;
#pragma unroll
  for (int i = 0; i < AT; ++i)
    if (i < A) {
      const float w = W3[(n2c + i) * WS + t], wt = W3t[(n2c + i) * WS + t];
      pm = fmaf(a[i], w, pm); pb = fmaf(ab[i], w, pb); pt = fmaf(at[i], wt, pt);
    }
  const float wqv = wq[t], wqtv = wqt[t];
  const float h3b = fmaxf(pb, 0.f);
  const float dzm = pm > 0.f ? wqv : 0.f;               // dz of the concat layer on the 2nd evaluation (dz of q is 1)
  if (rv && t < n3) h.h3_out[(long)row * h.ld_h3 + t] = h3b;
  HCK();
  const float qb = team_sum(h3b * wqv) + wq[N3P], qt = team_sum(fmaxf(pt, 0.f) * wqtv) + wqt[N3P];
#pragma unroll
  for (int i = 0; i < AT; ++i)
    if (i < A) { dqda[i] = team_sum(dzm * W3[(n2c + i) * WS + t]); adz[i] = -dqda[i] * (1.f - a[i] * a[i]); }
  const float td = rv ? qb - (rrow + (mrow * h.discount) * qt) : 0.f;
  const float dzq = td * (2.f / (float)h.B);
  if (rv && t == 0) { h.td[row] = td; h.dzq[row] = dzq; h.q_out[row] = qb; h.tq_out[row] = qt; }
  if (rv && t < A) {
#pragma unroll
    for (int i = 0; i < AT; ++i)
      if (i == t) {
        h.a_out[(long)row * A + i] = a[i]; h.dq_da[(long)row * A + i] = dqda[i]; h.adz[(long)row * A + i] = adz[i];
        h.cat_splice[(long)row * h.ld_h2c + i] = ab[i];
      }
  }
  HCK();
  // ---- one layer back: the actor's last hidden layer, the critic's concat layer and the layer feeding it
  float dz2 = 0.f;
  if (t < n2a) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AT; ++i)
      if (i < A) s = fmaf(adz[i], Wo[t * A + i], s);
    dz2 = xav > 0.f ? (h.relu_x2 ? 2.f * s : s) : 0.f;
    if (rv) h.dz_h2a[(long)row * n2a + t] = dz2;
  }
  if (n1a) sc2[t] = dz2;
  const float d3 = h3b > 0.f ? dzq * wqv : 0.f;
  if (rv && t < n3) h.dz3[(long)row * n3 + t] = d3;
  sc3[t] = d3;                                          // (a team is one wave: the reads below see it after the fence)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  HCK();
  if (rv && t < n2c) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // units >= n3: dz3 is zero there, the weights finite
#pragma unroll
    for (int j = 0; j < N3P; j += 4) {
      const heads_f4 dv = *reinterpret_cast<const heads_f4*>(sc3 + j), wv4 = ld4(W3 + t * WS + j);
      s0 = fmaf(dv[0], wv4[0], s0); s1 = fmaf(dv[1], wv4[1], s1); s2 = fmaf(dv[2], wv4[2], s2); s3 = fmaf(dv[3], wv4[3], s3);
    }
    h.dz2c[(long)row * n2c + t] = xcv > 0.f ? (s0 + s1) + (s2 + s3) : 0.f;
  }
  if (n1a && rv) {   // ---- and the actor layer below its last hidden layer
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const int j = t + n * HEADS_TEAM;
      if (j < n1a) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;    // units >= n2a: dz2 is zero there, the weights finite
#pragma unroll
        for (int c = 0; c < HEADS_TEAM; c += 4) {
          const heads_f4 dv = *reinterpret_cast<const heads_f4*>(sc2 + c), wv4 = ld4(W2 + j * n2a + c);
          s0 = fmaf(dv[0], wv4[0], s0); s1 = fmaf(dv[1], wv4[1], s1); s2 = fmaf(dv[2], wv4[2], s2); s3 = fmaf(dv[3], wv4[3], s3);
        }
        h.dz_h1a[(long)row * n1a + j] = x1v[n] > 0.f ? (s0 + s1) + (s2 + s3) : 0.f;
      }
    }
  }
  HCK();
  // ---- loss = mean(td^2): one partial per workgroup; the reader adds the partials in order (cpp_ddpg_last_stats)
  __shared__ double lred[HEADS_ROWS];
  if (t == 0) lred[team] = (double)td * (double)td;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < HEADS_ROWS; ++i) s += lred[i];
    h.loss_part[blockIdx.x] = s;
  }
#ifdef HEADS_CLOCK
  HCK();
  if (tid == 0 && blockIdx.x == 3) printf("HEADSCLK %llu %llu %llu %llu %llu %llu %llu (10 ns ticks)\n", ck[1]-ck[0], ck[2]-ck[1], ck[3]-ck[2], ck[4]-ck[3], ck[5]-ck[4], ck[6]-ck[5], ck[7]-ck[6]);
#endif
}

size_t ddpg_heads_lds_bytes(const DdpgHeadsArgs& h) {
  const size_t k3 = h.n2c + h.A + 1, n3p = HEADS_N3P, n2cp = (h.n2c + 3) & ~3;
  const size_t wfl = (k3 * h.n3 + HEADS_WSLACK + 3) & ~(size_t)3;
  const size_t w2fl = h.n1a ? (((size_t)(h.n1a + 1) * h.n2a + HEADS_WSLACK + 3) & ~(size_t)3) : 0, n1ap = (h.n1a + 3) & ~3;
  const size_t w = ((2 * wfl + 2 * (n3p + 4) + 2 * (size_t)(h.n2a + 1) * h.A + 3) & ~(size_t)3) + 2 * w2fl;
  const size_t f = w + (size_t)HEADS_ROWS * (n3p + 2 * n2cp + ((2 * h.n2a + 3) & ~3) + (h.n1a ? HEADS_TEAM + 2 * n1ap : 0));
  return f * sizeof(float);
}

bool ddpg_heads_supported(const DdpgHeadsArgs& h) {
  if (h.n1a && !(h.n1a <= HEADS_N1MAX && (h.n2a & 1) == 0 &&
                 (h.n1a + 1) * h.n2a + HEADS_WSLACK + 4 <= 4 * HEADS_NW4P * HEADS_THREADS)) return false;
  return h.A <= HEADS_AMAX && h.n3 <= HEADS_N3P && h.n2a <= HEADS_TEAM && h.n2c <= HEADS_TEAM &&
         (h.n2a + 1) * h.A <= 2 * HEADS_THREADS && (h.n2c + h.A + 1) * h.n3 + HEADS_WSLACK + 4 <= 4 * HEADS_NW4 * HEADS_THREADS && (h.n3 & 1) == 0 && ddpg_heads_lds_bytes(h) <= 120 * 1024 && (h.B + HEADS_ROWS - 1) / HEADS_ROWS <= DDPG_HEADS_MAX_WGS;
}

int launch_ddpg_heads(cpp_ctx* ctx, const DdpgHeadsArgs& h) {
  const size_t lds = ddpg_heads_lds_bytes(h);
  typedef void (*kern_t)(const DdpgHeadsArgs);
  static const kern_t kerns[6] = {ddpg_heads_kernel<1, true>, ddpg_heads_kernel<2, true>, ddpg_heads_kernel<4, true>,
                                  ddpg_heads_kernel<8, true>, ddpg_heads_kernel<4, false>, ddpg_heads_kernel<8, false>};
  const int ki = h.A == 1 ? 0 : h.A == 2 ? 1 : h.A == 4 ? 2 : h.A == 8 ? 3 : h.A == 3 ? 4 : 5;
  static size_t attr[CPP_MAX_DEVICES][6] = {};      // (kernel attributes are per device)
  size_t& have = attr[cpp_dev_slot(ctx)][ki];
  if (lds > have) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
  }
  prof_begin(ctx);
  hipLaunchKernelGGL(kerns[ki], dim3((h.B + HEADS_ROWS - 1) / HEADS_ROWS), dim3(HEADS_THREADS), lds, ctx->stream, h);
  LAUNCH_CHECK();
  prof_end(ctx, K_HEADS);
  return 0;
}

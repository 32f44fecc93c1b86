// DDPG "core" kernel: everything between the first fully connected layers' activations of the four networks and the ONE backward
// GEMM level that is left, row-local, on the matrix pipes (v_mfma_f32_16x16x4_f32) -- the reference's stacks (ddpg_cartpole.py:95-100,
// :166-171: actor 3 hidden layers + tanh head, critic hidden1 -> hidden2 -> [hidden2, action] -> hidden3 -> q).  It replaces one forward
// GEMM level, ddpg_heads_kernel and the dX half of one backward level (naf_mlp_kernel's scheme, gemm.hip, on twelve layers).
//
// A workgroup owns 16 batch rows (grid = ceil(B / 16)), its four waves one 16-column tile of every layer each (tiles w, w + 4, ...
// where a layer has more than four).  A layer is: A operand = the previous activations of the 16 rows in LDS (accumulator layout ->
// A layout goes through LDS), B operand = the weight matrix straight from global memory through a bounded buffer descriptor (a k or a
// column outside the matrix asks behind its end: zero, no branch), requested one stage ahead of its use.
//   forward   h1a = relu(h0a W1a), h2a = relu(h1a W2a), a = tanh(h2a Wo)      (live and target actor)
//             h1c = relu(h0c W1c), p = h1c W3[:n1c] + b3 + action W3[n1c:]     (live critic on mu(s1) and on the fed action; target on mu'(s2))
//             q = relu(p_fed) wq + bq, q' likewise, dQ/da = (relu'(p_mu) wq) W3[n1c:]^T
//   TD        td = q - (r + mask discount q'), loss partial, dz_q = 2 td / B, grad_ys = -dQ/da (1 - a^2)
//   backward  dz3 = relu'(p_fed) dz_q wq, dz1c = relu'(h1c)(dz3 W3[:n1c]^T), dz0c = relu'(h0c)(dz1c W1c^T)
//             dz2a = relu'(h2a)(grad_ys Wo^T), dz1a = relu'(h1a)(dz2a W2a^T), dz0a = relu'(h0a)(dz1a W1a^T)
// The loss is left as one partial per workgroup (cpp_ddpg_last_stats adds them in order, as for ddpg_heads_kernel).
#include "common.h"

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
namespace {
constexpr int OOB = 0x7FFFFF00;
constexpr int SA = 26, SC = 51, SH = 13;              // k steps: 104 (actor layers), 204 (critic layer 1), 52 (the 50-wide layers)
constexpr int LA = 105, LC = 205, LH = 53;            // odd LDS row strides: the A-operand reads of 16 rows fall on different banks
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t desc(const float* p, long floats) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000); }
__device__ __forceinline__ float ldf(const rsrc_t& r, int off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
// B operand of S k steps: b[s] = M[k * sk + n * sn] for k = 4 s + lj < K, n < N
template <int S> __device__ __forceinline__ void ldB(float (&b)[S], const rsrc_t& r, int sk, int sn, int K, int N, int n, int lj) {
#pragma unroll
  for (int s = 0; s < S; ++s) { const int k = 4 * s + lj; b[s] = ldf(r, (k < K && n < N) ? (k * sk + n * sn) * 4 : OOB); }
}
template <int S> __device__ __forceinline__ f32x4 mm(const float* As, const float (&b)[S], int lj) {
  f32x4 c = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < S; ++s) c = MFMA16(As[4 * s + lj], b[s], c);
  return c;
}
}

size_t ddpg_core_lds_bytes() { return (size_t)(2 * 16 * LA + 2 * 16 * LC + 2 * 16 * LA + 10 * 16 * LH + 16 * LA + 16 * 8 * 4) * sizeof(float); }

// 8 waves: a stage's (layer, network, column tile) tasks are dealt round robin to the waves (task t -> wave t mod 8), longest tasks
// first (a wave's outstanding-load counter ends at 63: the first version, four waves with up to 250 loads each, spent 9 of its 36 us
// issuing them; sixteen waves leave 128 registers per lane, not enough for two tasks' B operands).  The ReLU masks of the backward
// stages come from the activations in LDS, so a backward tile need not run on the wave that produced the forward tile.
constexpr int CORE_WAVES = 8, CORE_THREADS = 64 * CORE_WAVES;
__global__ __launch_bounds__(CORE_THREADS) void ddpg_core_kernel(const DdpgCoreArgs g) {
  extern __shared__ __attribute__((aligned(16))) float cl[];
#ifdef CORE_CLOCK
  unsigned long long ck[20]; int nck = 0;
#define CCK() ck[nck++] = __builtin_amdgcn_s_memrealtime()
#else
#define CCK()
#endif
  CCK();
  float* x0a = cl;               float* x0ta = x0a + 16 * LA;
  float* x0c = x0ta + 16 * LA;   float* x0tc = x0c + 16 * LC;
  float* h1a = x0tc + 16 * LC;   float* h1ta = h1a + 16 * LA;
  float* h2a = h1ta + 16 * LA;   float* h2ta = h2a + 16 * LH;
  float* h1c = h2ta + 16 * LH;   float* h1tc = h1c + 16 * LH;
  float* h3b = h1tc + 16 * LH;   float* h3t = h3b + 16 * LH;   float* dz3m = h3t + 16 * LH;
  float* dz3s = dz3m + 16 * LH;  float* dz1cs = dz3s + 16 * LH; float* dz2as = dz1cs + 16 * LH;
  float* dz1as = dz2as + 16 * LH;
  float* sm = dz1as + 16 * LA;   // small per-row values: [16][8] x 4: actions (live), actions (target), (dqda, grad_ys), (q, q', dzq, -)
  float* acts = sm;              float* actt = sm + 128;       float* dqs = sm + 256;       float* qs = sm + 384;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lj = lane >> 4, r0 = blockIdx.x * 16;
  const int A = g.A, n0a = g.n0a, n1a = g.n1a, n2a = g.n2a, n0c = g.n0c, n1c = g.n1c, n3 = g.n3;
  const int T1a = (n1a + 15) >> 4, T1c = (n1c + 15) >> 4, T2a = (n2a + 15) >> 4, T3 = (n3 + 15) >> 4, T0a = (n0a + 15) >> 4, T0c = (n0c + 15) >> 4;
  // ---- descriptors, built where they are used (twenty of them alive at once spill the scalar registers)
#define rh0a desc(g.h0a, (long)g.B * g.ld0a)
#define rh0ta desc(g.h0ta, (long)g.B * g.ld0a)
#define rh0c desc(g.h0c, (long)g.B * g.ld0c)
#define rh0tc desc(g.h0tc, (long)g.B * g.ld0c)
#define rW1a desc(g.W1a, (long)(n0a + 1) * n1a)
#define rW1ta desc(g.W1ta, (long)(n0a + 1) * n1a)
#define rW2a desc(g.W2a, (long)(n1a + 1) * n2a)
#define rW2ta desc(g.W2ta, (long)(n1a + 1) * n2a)
#define rWoa desc(g.Woa, (long)(n2a + 1) * A)
#define rWota desc(g.Wota, (long)(n2a + 1) * A)
#define rW1c desc(g.W1c, (long)(n0c + 1) * n1c)
#define rW1tc desc(g.W1tc, (long)(n0c + 1) * n1c)
#define rW3 desc(g.W3, (long)(n1c + A + 1) * n3)
#define rW3t desc(g.W3t, (long)(n1c + A + 1) * n3)
#define rwq desc(g.wq, n3 + 1)
#define rwqt desc(g.wqt, n3 + 1)
#define rW3act desc(g.W3 + (long)n1c * n3, (long)A * n3)
#define ract desc(g.act, (long)g.B * A)
#define rr_ desc(g.r, g.B)
#define rm_ desc(g.mask, g.B)
  // ---- stage 0: the rows of h0 (four networks) into LDS, coalesced
  constexpr int XA = 4, XC = 7;                              // ceil(16 * 104 / 512), ceil(16 * 204 / 512) elements per thread
  float ga[XA], gta[XA], gc[XC], gtc[XC];
  const int K0a = n0a + 1, K0c = n0c + 1;
#pragma unroll
  for (int u = 0; u < XA; ++u) {
    const int e = tid + CORE_THREADS * u, rr = e / K0a, k = e - rr * K0a, o = rr < 16 ? ((r0 + rr) * g.ld0a + k) * 4 : OOB;
    ga[u] = ldf(rh0a, o); gta[u] = ldf(rh0ta, o);
  }
#pragma unroll
  for (int u = 0; u < XC; ++u) {
    const int e = tid + CORE_THREADS * u, rr = e / K0c, k = e - rr * K0c, o = rr < 16 ? ((r0 + rr) * g.ld0c + k) * 4 : OOB;
    gc[u] = ldf(rh0c, o); gtc[u] = ldf(rh0tc, o);
  }
  // stage A's B operands: tasks [critic layer 1: live tiles, target tiles][actor layer 1: live tiles, target tiles] (the 51-step ones first:
  // a wave's second and third task are 26-step ones)
  const int NTA = 2 * T1a + 2 * T1c;
  float bA0[SC], bA1[SA], bA2[SA];
  int kindA[3], netA[3], tileA[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    int t = w + CORE_WAVES * q;
    kindA[q] = -1; netA[q] = 0; tileA[q] = 0;
    if (t < NTA) {                                           // (uniform per wave)
      if (t < 2 * T1c) { kindA[q] = 1; netA[q] = t / T1c; tileA[q] = t - netA[q] * T1c; }
      else { t -= 2 * T1c; kindA[q] = 0; netA[q] = t / T1a; tileA[q] = t - netA[q] * T1a; }
    }
  }
  if (kindA[0] == 1) ldB<SC>(bA0, netA[0] ? rW1tc : rW1c, n1c, 1, K0c, n1c, 16 * tileA[0] + li, lj);
  else if (kindA[0] == 0) { float (&b)[SA] = *reinterpret_cast<float (*)[SA]>(&bA0[0]); ldB<SA>(b, netA[0] ? rW1ta : rW1a, n1a, 1, K0a, n1a, 16 * tileA[0] + li, lj); }
  if (kindA[1] == 0) ldB<SA>(bA1, netA[1] ? rW1ta : rW1a, n1a, 1, K0a, n1a, 16 * tileA[1] + li, lj);
  if (kindA[2] == 0) ldB<SA>(bA2, netA[2] ? rW1ta : rW1a, n1a, 1, K0a, n1a, 16 * tileA[2] + li, lj);
  // the scalar stage's inputs (wave 0, one lane per row) and the fed actions in the accumulator layout (the concat layer's live tiles)
  const int srow = r0 + li;
  const bool srv = w == 0 && lj == 0 && srow < g.B;
  float fact[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fact[i] = ldf(ract, (srv && i < A) ? (srow * A + i) * 4 : OOB);
  const float rrow = ldf(rr_, srv ? srow * 4 : OOB), mrow = ldf(rm_, srv ? srow * 4 : OOB);
  // LDS: zeros and the bias inputs (column n of a buffer whose next layer has a bias row is 1.0)
  for (int i = tid; i < 16 * LA; i += CORE_THREADS) {
    const int c = i % LA;
    h1a[i] = c == n1a ? 1.f : 0.f; h1ta[i] = c == n1a ? 1.f : 0.f; dz1as[i] = 0.f; x0a[i] = 0.f; x0ta[i] = 0.f;
  }
  for (int i = tid; i < 16 * LC; i += CORE_THREADS) { x0c[i] = 0.f; x0tc[i] = 0.f; }
  for (int i = tid; i < 16 * LH; i += CORE_THREADS) {
    const int c = i % LH;
    h2a[i] = c == n2a ? 1.f : 0.f; h2ta[i] = c == n2a ? 1.f : 0.f;
    h1c[i] = 0.f; h1tc[i] = 0.f; h3b[i] = 0.f; h3t[i] = 0.f; dz3m[i] = 0.f; dz3s[i] = 0.f; dz1cs[i] = 0.f; dz2as[i] = 0.f;
  }
  if (tid < 512) sm[tid] = 0.f;
  CCK();
  __syncthreads();
#pragma unroll
  for (int u = 0; u < XA; ++u) {
    const int e = tid + CORE_THREADS * u, rr = e / K0a, k = e - rr * K0a;
    if (rr < 16) { x0a[rr * LA + k] = ga[u]; x0ta[rr * LA + k] = gta[u]; }
  }
#pragma unroll
  for (int u = 0; u < XC; ++u) {
    const int e = tid + CORE_THREADS * u, rr = e / K0c, k = e - rr * K0c;
    if (rr < 16) { x0c[rr * LC + k] = gc[u]; x0tc[rr * LC + k] = gtc[u]; }
  }
  // stage B's B operands (two tasks per wave): [actor layer 2: live, target][concat layer, state part: live, target]
  const int NTB = 2 * T2a + 2 * T3;
  int kindB[2], netB[2], tileB[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int t = w + CORE_WAVES * q;
    kindB[q] = -1; netB[q] = 0; tileB[q] = 0;
    if (t < NTB) {
      if (t < 2 * T2a) { kindB[q] = 0; netB[q] = t / T2a; tileB[q] = t - netB[q] * T2a; }
      else { t -= 2 * T2a; kindB[q] = 1; netB[q] = t / T3; tileB[q] = t - netB[q] * T3; }
    }
  }
  float bB[2][SA];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = 16 * tileB[q] + li;
    if (kindB[q] == 0) ldB<SA>(bB[q], netB[q] ? rW2ta : rW2a, n2a, 1, n1a + 1, n2a, n, lj);
    else if (kindB[q] == 1) { float (&b)[SH] = *reinterpret_cast<float (*)[SH]>(&bB[q][0]); ldB<SH>(b, netB[q] ? rW3t : rW3, n3, 1, n1c, n3, n, lj); }
  }
  // the concat layer's action rows, bias row and the q weights of this lane's unit (stage D runs on the waves of stage B's concat tiles;
  // a wave has at most ONE concat task: slot qc)
  const int qc = kindB[0] == 1 ? 0 : kindB[1] == 1 ? 1 : -1;
  const int netC = qc >= 0 ? netB[qc] : 0, nC = qc >= 0 ? 16 * tileB[qc] + li : 0;
  float w3a[4] = {0.f, 0.f, 0.f, 0.f}, b3v = 0.f, wqv = 0.f, fa[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) fa[i][k] = 0.f;
  if (qc >= 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) w3a[i] = ldf(netC ? rW3t : rW3, (i < A && nC < n3) ? ((n1c + i) * n3 + nC) * 4 : OOB);
    b3v = ldf(netC ? rW3t : rW3, nC < n3 ? ((n1c + A) * n3 + nC) * 4 : OOB);
    wqv = ldf(rwq, nC < n3 ? nC * 4 : OOB);
    if (netC == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) fa[i][k] = ldf(ract, (r0 + 4 * lj + i < g.B && k < A) ? ((r0 + 4 * lj + i) * A + k) * 4 : OOB);
    }
  }
  CCK();
  __syncthreads();
  // ---- stage A: the first layers of this kernel
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (kindA[q] < 0) continue;
    const int n = 16 * tileA[q] + li;
    f32x4 c;
    if (q == 0) {
      if (kindA[0] == 1) c = mm<SC>((netA[0] ? x0tc : x0c) + li * LC, bA0, lj);
      else { float (&b)[SA] = *reinterpret_cast<float (*)[SA]>(&bA0[0]); c = mm<SA>((netA[0] ? x0ta : x0a) + li * LA, b, lj); }
    } else c = mm<SA>((netA[q] ? x0ta : x0a) + li * LA, q == 1 ? bA1 : bA2, lj);
    float* dst = kindA[q] == 0 ? (netA[q] ? h1ta : h1a) : (netA[q] ? h1tc : h1c);
    const int ldd = kindA[q] == 0 ? LA : LH, nmax = kindA[q] == 0 ? n1a : n1c;
    float* gout = kindA[q] == 0 ? g.h1a_w : g.h1c_w;
    const int ldg = kindA[q] == 0 ? g.ld1a : g.ld1c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      const float v = fmaxf(c[i], 0.f);
      if (n < nmax) {
        dst[row * ldd + n] = v;
        if (netA[q] == 0 && r0 + row < g.B) gout[(long)(r0 + row) * ldg + n] = v;
      }
    }
  }
  // next: the actors' heads (waves 0 / 1), the q layers and dQ/da (waves 2 / 3 / 4)
  float bo[SH];
  if (w < 2) ldB<SH>(bo, w == 0 ? rWoa : rWota, A, 1, n2a + 1, A, li, lj);          // B[k][i] = Wo[k A + i]
  else if (w == 2) ldB<SH>(bo, rwq, 1, 0, n3, 1, li, lj);                            // B[k][0] = wq[k]
  else if (w == 3) ldB<SH>(bo, rwqt, 1, 0, n3, 1, li, lj);
  else if (w == 4) ldB<SH>(bo, rW3act, 1, n3, n3, A, li, lj);                        // B[k][i] = W3[(n1c + i) n3 + k]
  const float bqv = ldf(rwq, n3 * 4), bqtv = ldf(rwqt, n3 * 4);
  CCK();
  __syncthreads();
  // ---- stage B: actor layer 2; the concat layer's state part (its action part and bias are added when the actions exist)
  f32x4 pbase = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int n = 16 * tileB[q] + li;
    if (kindB[q] == 0) {
      const f32x4 c = mm<SA>((netB[q] ? h1ta : h1a) + li * LA, bB[q], lj);
      float* dst = netB[q] ? h2ta : h2a;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * lj + i;
        const float v = fmaxf(c[i], 0.f);
        if (n < n2a) { dst[row * LH + n] = v; if (netB[q] == 0 && r0 + row < g.B) g.h2a_w[(long)(r0 + row) * g.ld2a + n] = v; }
      }
    } else if (kindB[q] == 1) {
      float (&b)[SH] = *reinterpret_cast<float (*)[SH]>(&bB[q][0]);
      pbase = mm<SH>((netB[q] ? h1tc : h1c) + li * LH, b, lj);
    }
  }
  // next: stage H's B operands (dz1a: T1a tiles, dz1c: T1c tiles; up to two tasks per wave)
  const int NTH = T1c + T1a;
  int kindH[2], tileH[2];
  float bH[2][SH];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int t = w + CORE_WAVES * q;
    kindH[q] = t < T1c ? 0 : t < NTH ? 1 : -1; tileH[q] = kindH[q] == 0 ? t : t - T1c;
    const int n = 16 * tileH[q] + li;
    if (kindH[q] == 0) ldB<SH>(bH[q], rW3, 1, n3, n3, n1c, n, lj);                   // B[k = n][j] = W3[j n3 + n]
    else if (kindH[q] == 1) ldB<SH>(bH[q], rW2a, 1, n2a, n2a, n1a, n, lj);           // B[k = n2][j] = W2a[j n2a + n2]
  }
  CCK();
  __syncthreads();
  // ---- stage C: the actors' heads (wave 0: live, wave 1: target): a = tanh(h2a Wo)
  if (w < 2) {
    const f32x4 c = mm<SH>((w == 0 ? h2a : h2ta) + li * LH, bo, lj);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      if (li < A) {
        const float av = tanhf(c[i]);
        (w == 0 ? acts : actt)[row * 8 + li] = av;
        if (w == 0 && r0 + row < g.B) g.a_out[(long)(r0 + row) * A + li] = av;
      }
    }
  }
  CCK();
  __syncthreads();
  // ---- stage D: the pre-activations of the concat layer (live: on mu(s1) and on the fed action; target: on mu'(s2))
  if (qc >= 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      float pm = pbase[i] + b3v, pb = pm;
      const float* av = (netC ? actt : acts) + row * 8;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < A) { pm = fmaf(av[k], w3a[k], pm); pb = fmaf(fa[i][k], w3a[k], pb); }        // (uniform)
      if (nC < n3) {
        if (netC == 0) {
          const float hv = fmaxf(pb, 0.f);
          h3b[row * LH + nC] = hv; dz3m[row * LH + nC] = pm > 0.f ? wqv : 0.f;
          if (r0 + row < g.B) g.h3_w[(long)(r0 + row) * g.ld3 + nC] = hv;
        } else h3t[row * LH + nC] = fmaxf(pm, 0.f);
      }
    }
  }
  CCK();
  __syncthreads();
  // ---- stage E: q, q', dQ/da (waves 2, 3, 4)
  if (w >= 2 && w < 5) {
    const f32x4 c = mm<SH>((w == 2 ? h3b : w == 3 ? h3t : dz3m) + li * LH, bo, lj);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      if (w < 4) { if (li == 0) qs[row * 8 + (w - 2)] = c[i]; }
      else if (li < A) dqs[row * 8 + li] = c[i];
    }
  }
  // next: stage G's per-unit weights (waves 0 .. T3 - 1: wq; waves T3 .. T3 + T2a - 1: Wo's row)
  const int kindG = w < T3 ? 0 : w < T3 + T2a ? 1 : -1, tileG = kindG == 0 ? w : w - T3, nG = 16 * tileG + li;
  float wg[4] = {0.f, 0.f, 0.f, 0.f};
  if (kindG == 0) wg[0] = ldf(rwq, nG < n3 ? nG * 4 : OOB);
  else if (kindG == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wg[i] = ldf(rWoa, (i < A && nG < n2a) ? (nG * A + i) * 4 : OOB);
  }
  // ... and stage I's B operands (dz0a: T0a tiles first -- the 26-step ones --, dz0c: T0c tiles; up to three tasks per wave)
  const int NTI = T0c + T0a;
  float bI0[SA], bI1[SH], bI2[SH];
  int kindI[3], tileI[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int t = w + CORE_WAVES * q;
    kindI[q] = t < T0a ? 1 : t < NTI ? 0 : -1; tileI[q] = kindI[q] == 1 ? t : t - T0a;
  }
  if (kindI[0] == 1) ldB<SA>(bI0, rW1a, 1, n1a, n1a, n0a, 16 * tileI[0] + li, lj);                                                       // B[k = j][mm] = W1a[mm n1a + j]
  else if (kindI[0] == 0) { float (&b)[SH] = *reinterpret_cast<float (*)[SH]>(&bI0[0]); ldB<SH>(b, rW1c, 1, n1c, n1c, n0c, 16 * tileI[0] + li, lj); }   // B[k = j][mm] = W1c[mm n1c + j]
  if (kindI[1] == 0) ldB<SH>(bI1, rW1c, 1, n1c, n1c, n0c, 16 * tileI[1] + li, lj);
  if (kindI[2] == 0) ldB<SH>(bI2, rW1c, 1, n1c, n1c, n0c, 16 * tileI[2] + li, lj);
  CCK();
  __syncthreads();
  // ---- stage F: TD, loss, dz_q, grad_ys (wave 0, one lane per row)
  if (w == 0 && lj == 0) {
    const int row = li;
    const float qb = qs[row * 8 + 0] + bqv, qt = qs[row * 8 + 1] + bqtv;
    const float td = srv ? qb - (rrow + (mrow * g.discount) * qt) : 0.f;
    const float dzq = td * (2.f / (float)g.B);
    qs[row * 8 + 2] = dzq;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < A) {
        const float av = acts[row * 8 + i], dq = dqs[row * 8 + i];
        const float ad = -dq * (1.f - av * av);
        dqs[row * 8 + 4 + i] = ad;
        if (srv) { g.dq_da[(long)srow * A + i] = dq; g.adz[(long)srow * A + i] = ad; g.h1c_w[(long)srow * g.ld1c + n1c + i] = fact[i]; }
      }
    if (srv) { g.td[srow] = td; g.dzq[srow] = dzq; g.q_out[srow] = qb; g.tq_out[srow] = qt; }
    double s2 = (double)td * (double)td;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
    if (li == 0) g.loss_part[blockIdx.x] = s2;
  }
  CCK();
  __syncthreads();
  // ---- stage G: dz3 and dz2a (accumulator layout, no contraction longer than the action count)
  if (kindG >= 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      if (kindG == 0) {
        const float d3 = (nG < n3 && h3b[row * LH + nG] > 0.f) ? qs[row * 8 + 2] * wg[0] : 0.f;
        if (nG < n3) { dz3s[row * LH + nG] = d3; if (r0 + row < g.B) g.dz3[(long)(r0 + row) * n3 + nG] = d3; }
      } else {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < A) s = fmaf(dqs[row * 8 + 4 + k], wg[k], s);
        const float d2 = (nG < n2a && h2a[row * LH + nG] > 0.f) ? s : 0.f;
        if (nG < n2a) { dz2as[row * LH + nG] = d2; if (r0 + row < g.B) g.dz2a[(long)(r0 + row) * n2a + nG] = d2; }
      }
    }
  }
  CCK();
  __syncthreads();
  // ---- stage H: dz1c = relu'(h1c)(dz3 W3[:n1c]^T), dz1a = relu'(h1a)(dz2a W2a^T)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (kindH[q] < 0) continue;
    const int nH = 16 * tileH[q] + li;
    const f32x4 c = mm<SH>((kindH[q] == 0 ? dz3s : dz2as) + li * LH, bH[q], lj);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 4 * lj + i;
      if (kindH[q] == 0) {
        if (nH < n1c) { const float v = h1c[row * LH + nH] > 0.f ? c[i] : 0.f; dz1cs[row * LH + nH] = v; if (r0 + row < g.B) g.dz1c[(long)(r0 + row) * n1c + nH] = v; }
      } else {
        if (nH < n1a) { const float v = h1a[row * LA + nH] > 0.f ? c[i] : 0.f; dz1as[row * LA + nH] = v; if (r0 + row < g.B) g.dz1a[(long)(r0 + row) * n1a + nH] = v; }
      }
    }
  }
  CCK();
  __syncthreads();
  // ---- stage I: dz0c = relu'(h0c)(dz1c W1c^T), dz0a = relu'(h0a)(dz1a W1a^T)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (kindI[q] < 0) continue;
    const int mcol = 16 * tileI[q] + li;
    if (kindI[q] == 1) {                                       // (only a wave's first task can be an actor tile)
      const f32x4 c = mm<SA>(dz1as + li * LA, bI0, lj);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * lj + i;
        if (mcol < n0a && r0 + row < g.B) g.dz0a[(long)(r0 + row) * n0a + mcol] = x0a[row * LA + mcol] > 0.f ? c[i] : 0.f;
      }
    } else {
      f32x4 c;
      if (q == 0) { float (&b)[SH] = *reinterpret_cast<float (*)[SH]>(&bI0[0]); c = mm<SH>(dz1cs + li * LH, b, lj); }
      else c = mm<SH>(dz1cs + li * LH, q == 1 ? bI1 : bI2, lj);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 4 * lj + i;
        if (mcol < n0c && r0 + row < g.B) g.dz0c[(long)(r0 + row) * n0c + mcol] = x0c[row * LC + mcol] > 0.f ? c[i] : 0.f;
      }
    }
  }
#ifdef CORE_CLOCK
  CCK();
  if (tid == 0 && blockIdx.x == 3) { printf("CORECLK"); for (int i = 1; i < nck; ++i) printf(" %llu", ck[i] - ck[i - 1]); printf(" (10 ns ticks; %d stamps)\n", nck); }
#endif
}

bool ddpg_core_supported(const DdpgCoreArgs& g) {
  return g.A >= 1 && g.A <= 4 && g.n0a >= 1 && g.n0a + 1 <= 4 * SA && g.n1a >= 1 && g.n1a + 1 <= 4 * SA && g.n1a <= 128 && g.n2a >= 1 && g.n2a + 1 <= 4 * SH &&
         g.n0c >= 1 && g.n0c + 1 <= 4 * SC && g.n0c <= 256 && g.n1c >= 1 && g.n1c <= 4 * SH && g.n3 >= 1 && g.n3 <= 4 * SH && g.n0a <= 128 &&
         (g.B + 15) / 16 <= DDPG_HEADS_MAX_WGS && 16 * (g.n0a + 1) <= CORE_THREADS * 4 && 16 * (g.n0c + 1) <= CORE_THREADS * 7 &&
         2 * ((g.n1c + 15) / 16) <= CORE_WAVES && 2 * ((g.n1a + 15) / 16) + 2 * ((g.n1c + 15) / 16) <= 3 * CORE_WAVES &&
         2 * ((g.n2a + 15) / 16) <= CORE_WAVES && 2 * ((g.n2a + 15) / 16) + 2 * ((g.n3 + 15) / 16) <= 2 * CORE_WAVES && 2 * ((g.n3 + 15) / 16) <= CORE_WAVES &&
         (g.n1c + 15) / 16 + (g.n1a + 15) / 16 <= 2 * CORE_WAVES && (g.n0a + 15) / 16 <= CORE_WAVES && (g.n0c + 15) / 16 + (g.n0a + 15) / 16 <= 3 * CORE_WAVES &&
         (g.n3 + 15) / 16 + (g.n2a + 15) / 16 <= CORE_WAVES;
}

int launch_ddpg_core(cpp_ctx* ctx, const DdpgCoreArgs& g) {
  const size_t lds = ddpg_core_lds_bytes();
  static size_t attr[CPP_MAX_DEVICES] = {};
  size_t& have = attr[cpp_dev_slot(ctx)];
  if (lds > have) {
    HIP_CHECK(hipFuncSetAttribute((const void*)ddpg_core_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    have = lds;
  }
  prof_begin(ctx);
  hipLaunchKernelGGL(ddpg_core_kernel, dim3((g.B + 15) / 16), dim3(CORE_THREADS), lds, ctx->stream, g);
  LAUNCH_CHECK();
  prof_end(ctx, K_HEADS);
  return 0;
}

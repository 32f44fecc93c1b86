// conv1 forward on the f16 matrix pipes, ROW-STREAMING with the weights resident in registers ("rs16"): v_mfma_f32_16x16x32_f16.
//
// Same arithmetic contract as conv_k16.h (base_network.py:95-107: whitening folded into the weights chunk by chunk, raw f16 pixels as the
// other operand, two f16 pieces of W s, f32 accumulation, bias + ReLU + 2x2 max-pool + arg-max code in the epilogue) -- a different
// formulation of the same sums, built because conv_k16.h's launch is its waves' NON-MFMA instruction chain (per row and
// wave 100-150 VALU, 31 LDS reads of rotating weights, 26 waits, a pool transpose through LDS, against 48 MFMAs):
//
//   * The MFMA's 16 ROWS are the 10 filters (A operand = weights, lane (li, lj) holds filter li's 8 consecutive k of lane group lj),
//     its 16 COLUMNS are pixels (B operand = the raw image row).  The weights of all KS x NCH chunks x 2 pieces (30 A operands = 120
//     VGPRs at 13 .. 18 channels, 20 at 7 .. 12, 10 at 3 / 6) are loaded ONCE per wave from a prebuilt image and stay in registers: the
//     row loop reads no weights.
//   * The KS output rows an input row contributes to are KS ACCUMULATOR SETS (+ one that is being written out): a set restarts
//     through the C operand of its first MFMA (bias and border constant), nothing rotates, nothing is reset.
//   * The image rows go through LDS, coalesced: every lane loads 16 contiguous bytes of the strip's row segment two rows ahead of its
//     use, the segment sits in the wave's own LDS slot one row ahead, the two pixels of SAME padding are overwritten THERE with the
//     pivots, and the operand windows are 4-byte aligned dword reads at per-lane addresses (a misaligned ds_read_b128 is served at a
//     fraction of the rate).  An ODD channel count (9, 15: a pixel is an odd number of halves) keeps TWO copies of the segment, the second
//     displaced by one half (loaded from the image 2 bytes earlier): the even pixels' windows are aligned in the first, the odd
//     pixels' in the second.
//   * The two column tiles of a wave hold the EVEN and the ODD pixels of its 32-pixel strip: the x half of the 2x2 pool is an
//     element-wise max of the two tiles' accumulators, the y half is the previous output row's registers (same lanes).  The pooled
//     row crosses the wave's LDS slot once into row layout and leaves as five contiguous stores (16 / 8 / 8 / 8 / 4 bytes per lane:
//     f32, three bf16 planes, codes).
//   * 20 NCH MFMAs per input row and wave (10 of 16 rows are filters, against conv_k16.h's 50 of 64 columns), ~50 VALU, ~20 LDS
//     accesses, 2 .. 4 loads and the compiler's own waits beside them.  No hand-counted wait: every load is a builtin.
//
// The per-network operand image (pieces of W s 2^S, the chunks' ones weights, the border constants, the pivots) is built by
// conv1_image_kernel below -- conv_k16.h's per-workgroup setup, run ONCE per network and minibatch instead of in all 512 workgroups
// (in the fused steps: by a rider of the optimiser's launch, optim.hip).
#pragma once
#include "conv_k16.h"

template <int CIN, int NPCS = F16_PIECES>
struct Rs16Geom {
  static constexpr int KS = 5, P = 2, NO = KYO_NO, RK = 30;
  static constexpr int KROW = KS * CIN, NCH = (KROW + RK - 1) / RK, NPC = NPCS;
  static constexpr int NSET = KS + 1;                        // accumulator sets: KS in flight + the one whose pooled row is being written
  static constexpr int NRC = 2 * P + 1;                      // output row classes (rows 0, 1, interior, H - 2, H - 1)
  static constexpr int NUNIT = KS * NCH * 4 * 16;            // 16-byte operand units per piece: (ky, chunk, lane group g, row o of 16)
  static constexpr int UNIT_BYTES = KS * NCH * NPC * 1024;   // [ky][chunk][piece][g][o16] x 16 bytes: one A operand per 1 KB
  static constexpr int CT_OFF = UNIT_BYTES;                  // float ct[NRC][5][16]: a set's restart value, bias 2^S - E[rc][o][x class]
  static constexpr int CT_BYTES = NRC * 5 * 16 * 4;          //   (x class 0: not a border column; 1 .. 4: x = 0, 1, W - 2, W - 1)
  static constexpr int PW_OFF = CT_OFF + CT_BYTES;           // unsigned pw[chunk][lj][v]: what dword v of lane group lj's window is replaced by
  static constexpr int PW_BYTES = NCH * 16 * 4;              //   (the pivots p_c of its two k; the ones slots' 2^T)
  static constexpr int SC_OFF = PW_OFF + PW_BYTES;           // float inv = 2^-S
  static constexpr int REC_BYTES = (SC_OFF + 16 + 255) & ~255;
};

// ---- the operand image of one network: conv_k16.h's setup, value for value (same f32 / f64 operations in the same order), written
// once to global memory.  One workgroup per network.
struct Conv1ImageArgs {
  const float* w; const float* bias; const float* scale; const float* shift; float wscale; int nout; unsigned char* rec;
  // optional (the builder riding in the optimiser's launch, optim.hip): the weights are the ones THIS launch's SGD update is about to
  // write, w - lr * (gw * gscale), recomputed with the update's own operations -- the builder waits for nobody
  const float* gw; const float* gb; float lr, gscale;
  float* mw; float* mb; float momentum;      // OPT_MOMENTUM: the parameters' accumulator slots (read, advanced and written back here)
  float* w_out; float* b_out;     // ... and written by THIS workgroup (the update's own workgroups leave these parameters alone: no one reads a half-updated tensor)
};
struct Conv1ImageArgsN { Conv1ImageArgs a[CONV_BATCH_MAX]; int n; };

// pre(): called once the loads of the weights (gradients, accumulators) are in flight, returns the gradient scale of the update that
// rides here -- optim.hip's rider adds up the clipping norm and fetches the whitening table in there, under the loads' latency
struct Conv1ImageNoPre { __device__ __forceinline__ float operator()(float gscale) const { return gscale; } };
template <int CIN, int NPCS = F16_PIECES, typename Pre = Conv1ImageNoPre>
__device__ __forceinline__ void conv1_image_body(const Conv1ImageArgs& a, unsigned char* lds_raw, Pre pre = Pre()) {
  typedef Rs16Geom<CIN, NPCS> G;
  constexpr int KS = G::KS, P = G::P, NO = G::NO, NCH = G::NCH, NPC = G::NPC, RK = G::RK;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nout = a.nout;
  // LDS: the record itself, then scratch
  unsigned char* rec = lds_raw;
  double* mu = reinterpret_cast<double*>(lds_raw + G::REC_BYTES);           // [CIN (+1)]
  constexpr int NU = KS * NCH * 4 * NO;                                     // conv_k16.h's unit index: o fastest (NO), then g, chunk, ky
  double* opart = mu + ((CIN + 1) & ~1);                                    // [NU]
  constexpr int NB = 7 / CIN + 2;
  float* bpart = reinterpret_cast<float*>(opart + NU);                      // [NU][NB]
  float* scl = bpart + NU * NB;                                             // [CIN]
  float* dmu = scl + CIN;                                                   // [CIN]
  float* red = dmu + CIN;                                                   // [8]
  float* ctab = red + 8;                                                    // E[NRC][NO][4]
  unsigned short* pivot = reinterpret_cast<unsigned short*>(ctab + G::NRC * NO * 4);      // [CIN]
  constexpr int NUW = (NU + CONV_THREADS - 1) / CONV_THREADS;
  // the layer's weights (as this launch's SGD update leaves them, if it rides there) through LDS: consecutive threads load consecutive
  // floats -- a thread fetching its own 24 (ky, k, o) values asked for 40-byte strides, 2.6 us of this workgroup (6.5 with the gradients)
  float* wst = reinterpret_cast<float*>(lds_raw);              // [KS * KROW * nout] -- in the record's place, which is written later
  const int nwts = KS * G::KROW * nout;
  float gscale_b = a.gscale;
  {
    constexpr int NWT = (KS * G::KROW * NO + CONV_THREADS - 1) / CONV_THREADS;      // (every load in flight before the first use)
    float wr[NWT], gr[NWT];
#pragma unroll
    for (int n = 0; n < NWT; ++n) { const int i = tid + n * CONV_THREADS; wr[n] = a.w[i < nwts ? i : 0]; }
    float mr[NWT];
    if (a.gw) {
#pragma unroll
      for (int n = 0; n < NWT; ++n) { const int i = tid + n * CONV_THREADS; gr[n] = a.gw[i < nwts ? i : 0]; }
      if (a.mw) {
#pragma unroll
        for (int n = 0; n < NWT; ++n) { const int i = tid + n * CONV_THREADS; mr[n] = a.mw[i < nwts ? i : 0]; }
      }
    }
    const float gscale = pre(a.gscale);
    gscale_b = gscale;
#pragma unroll
    for (int n = 0; n < NWT; ++n) {
      const int i = tid + n * CONV_THREADS;
      if (i < nwts) {
        float w = wr[n];
        if (a.gw && a.mw) {                                    // (opt_apply_kernel's own expressions)
          const float acc = momentum_accum(mr[n], gr[n], gscale, a.momentum);
          a.mw[i] = acc;
          w = momentum_step(w, acc, a.lr);
        } else if (a.gw) w = sgd_update(w, gr[n], gscale, a.lr);
        wst[i] = w;
        if (a.gw && a.w_out) a.w_out[i] = w;
      }
    }
  }
  __syncthreads();
  float wv[NUW][8];
  float vmax = 0.f;
#pragma unroll
  for (int n = 0; n < NUW; ++n) {
    const int u = tid + n * CONV_THREADS;
    const int o = u % NO, g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH, ky = u / (4 * NO * NCH);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kl = 8 * g + e, k = RK * ch + kl;
      const bool real = u < NU && o < nout && kl < RK && k < G::KROW;
      wv[n][e] = real ? wst[(ky * G::KROW + k) * nout + o] : 0.f;
    }
  }
  float* biasn = reinterpret_cast<float*>(pivot + ((CIN + 7) & ~7));      // [16] the layer's biases as this launch leaves them
  if (tid < 16) {
    float bo = (a.bias && tid < nout) ? a.bias[tid] : 0.f;
    if (a.bias && a.gb && tid < nout) {
      if (a.mb) { const float acc = momentum_accum(a.mb[tid], a.gb[tid], gscale_b, a.momentum); a.mb[tid] = acc; bo = momentum_step(bo, acc, a.lr); }
      else bo = sgd_update(bo, a.gb[tid], gscale_b, a.lr);
      if (a.b_out) a.b_out[tid] = bo;
    }
    biasn[tid] = bo;
  }
  if (tid < CIN) {
    const float s_c = a.scale[tid], t_c = a.shift[tid];
    const double m_c = s_c != 0.f ? -(double)t_c / (double)s_c : 0.0;
    const _Float16 p_c = (_Float16)(float)m_c;
    mu[tid] = m_c;
    pivot[tid] = __builtin_bit_cast(unsigned short, p_c);
    scl[tid] = s_c;
    dmu[tid] = (float)((double)(float)p_c - m_c);
  }
  __syncthreads();
#pragma unroll
  for (int n = 0; n < NUW; ++n) {
    const int u = tid + n * CONV_THREADS;
    const int g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = wv[n][e];
      v *= scl[(RK * ch + 8 * g + e) % CIN];
      if (a.wscale != 0.f) v *= a.wscale;
      wv[n][e] = v;
      vmax = fmaxf(vmax, fabsf(v));
    }
  }
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
  if (lane == 0) red[wave] = vmax;
  __syncthreads();
  vmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int i = tid; i < G::REC_BYTES / 16; i += CONV_THREADS) reinterpret_cast<k16_u32x4*>(rec)[i] = (k16_u32x4){0u, 0u, 0u, 0u};
  __syncthreads();                                   // (every thread has its weights in registers: the record takes the staging area over)
  int S = 0;
  if (vmax > 0.f && vmax < 3.0e38f) S = 14 - ilogbf(vmax);
  S = S > 100 ? 100 : (S < -100 ? -100 : S);
  const float sc = ldexpf(1.f, S), inv = ldexpf(1.f, -S);
#pragma unroll
  for (int n = 0; n < NUW; ++n) {
    const int u = tid + n * CONV_THREADS;
    const int o = u % NO, g = (u / NO) & 3, ch = (u / (4 * NO)) % NCH, ky = u / (4 * NO * NCH);
    unsigned short pcs[3][8];
    double po = 0.0;
    float pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) pb[b] = 0.f;
    const int kx0 = (RK * ch + 8 * g) / CIN;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = wv[n][e] * sc;
      const _Float16 hh = (_Float16)v;
      const float r1 = v - (float)hh;
      const _Float16 mm = (_Float16)r1;
      const float r2 = r1 - (float)mm;
      const _Float16 ll = (_Float16)r2;
      pcs[0][e] = __builtin_bit_cast(unsigned short, hh); pcs[1][e] = __builtin_bit_cast(unsigned short, mm); pcs[2][e] = __builtin_bit_cast(unsigned short, ll);
      const int kl = 8 * g + e, k = RK * ch + kl;
      if (kl < RK && k < G::KROW) {
        const double vp = NPC > 2 ? ((double)(float)hh + (double)(float)mm) + (double)(float)ll : (double)(float)hh + (double)(float)mm;
        const int c = k % CIN;
        po -= vp * mu[c];
        const float d = (float)vp * dmu[c];
        const int b = k / CIN - kx0;
#pragma unroll
        for (int bb = 0; bb < NB; ++bb) pb[bb] += bb == b ? d : 0.f;
      }
    }
    if (u < NU) {
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc)
        *reinterpret_cast<k16_u32x4*>(rec + ((ky * NCH + ch) * NPC + pc) * 1024 + (g * 16 + o) * 16) =
            (k16_u32x4){(unsigned)pcs[pc][0] | ((unsigned)pcs[pc][1] << 16), (unsigned)pcs[pc][2] | ((unsigned)pcs[pc][3] << 16),
                        (unsigned)pcs[pc][4] | ((unsigned)pcs[pc][5] << 16), (unsigned)pcs[pc][6] | ((unsigned)pcs[pc][7] << 16)};
      opart[u] = po;
#pragma unroll
      for (int b = 0; b < NB; ++b) bpart[NB * u + b] = pb[b];
    }
  }
  __syncthreads();
  constexpr int NJ1 = KS * NCH * NO;
  constexpr int NJW = (NJ1 + CONV_THREADS - 1) / CONV_THREADS;
  double osumv[NJW];
  double omax = 0.0;
#pragma unroll
  for (int jk = 0; jk < NJW; ++jk) {
    const int job = tid + jk * CONV_THREADS;
    osumv[jk] = 0.0;
    if (job < NJ1) {
      const int o = job % NO, cy = job / NO;
      const double* pp = opart + (cy * 4) * NO + o;
      const double acc = ((pp[0] + pp[NO]) + pp[2 * NO]) + pp[3 * NO];
      osumv[jk] = acc;
      omax = fmax(omax, fabs(acc));
    }
  }
  for (int job = tid; job < G::NRC * NO * 4; job += CONV_THREADS) {
    const int xi = job & 3, o = (job >> 2) % NO, rc = job / (4 * NO);
    const int kxlo = xi == 0 ? 0 : (xi == 1 ? 0 : (xi == 2 ? KS - 1 : KS - 2)), kxhi = xi == 0 ? 1 : (xi == 1 ? 0 : KS - 1);
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < KS; ++ky) {
      const bool seen = rc < P ? ky >= P - rc : (rc > P ? ky <= KS - 1 - (rc - P) : true);
      float kacc = 0.f;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int kxa = (RK * ch + 8 * g) / CIN;
          const float* bp = bpart + NB * ((((ky * NCH + ch) * 4) + g) * NO + o);
#pragma unroll
          for (int b = 0; b < NB; ++b) kacc += (kxa + b >= kxlo && kxa + b <= kxhi) ? bp[b] : 0.f;
        }
      acc += seen ? kacc : 0.f;
    }
    ctab[job] = acc;
  }
  for (int o = 32; o > 0; o >>= 1) omax = fmax(omax, __shfl_xor(omax, o));
  if (lane == 0) red[4 + wave] = (float)omax;
  __syncthreads();
  const float om = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])) * 1.0001f;
  int onesT = (om > 0.f && om < 3.0e38f) ? ilogbf(om) - 14 : 0;
  onesT = onesT < 0 ? 0 : (onesT > 15 ? 15 : onesT);
#pragma unroll
  for (int jk = 0; jk < NJW; ++jk) {
    const int job = tid + jk * CONV_THREADS;
    if (job < NJ1) {
      const int o = job % NO, ch = (job / NO) % NCH, ky = job / (NO * NCH);
      double v = ldexp(osumv[jk], -onesT);
#pragma unroll
      for (int slot = 0; slot < 2; ++slot)
#pragma unroll
        for (int pc = 0; pc < NPC; ++pc) {
          const _Float16 hh = (_Float16)(float)v;
          v -= (double)(float)hh;
          *reinterpret_cast<unsigned short*>(rec + ((ky * NCH + ch) * NPC + pc) * 1024 + (3 * 16 + o) * 16 + (6 + slot) * 2) = __builtin_bit_cast(unsigned short, hh);
        }
    }
  }
  // restart values: bias 2^S minus the border constant of the column's class
  for (int job = tid; job < G::NRC * 5 * 16; job += CONV_THREADS) {
    const int o = job & 15, xs = (job >> 4) % 5, rc = job / 80;
    float v = 0.f;
    if (o < nout) {
      v = biasn[o] * sc;
      if (xs > 0) v -= ctab[(rc * NO + o) * 4 + xs - 1];
    }
    reinterpret_cast<float*>(rec + G::CT_OFF)[job] = v;
  }
  const unsigned ones_bits = (unsigned)(15 + onesT) << 10;
  for (int job = tid; job < NCH * 16; job += CONV_THREADS) {
    const int v = job & 3, lj = (job >> 2) & 3, ch = job >> 4;
    unsigned wd = 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kl = 8 * lj + 2 * v + h, k = RK * ch + kl;
      unsigned hv = 0u;
      if (kl >= RK) hv = ones_bits; else if (k < G::KROW) hv = pivot[k % CIN];
      wd |= hv << (16 * h);
    }
    reinterpret_cast<unsigned*>(rec + G::PW_OFF)[job] = wd;
  }
  if (tid == 0) *reinterpret_cast<float*>(rec + G::SC_OFF) = inv;
  __syncthreads();
  for (int i = tid; i < G::REC_BYTES / 16; i += CONV_THREADS) reinterpret_cast<k16_u32x4*>(a.rec)[i] = reinterpret_cast<const k16_u32x4*>(rec)[i];
}

template <int CIN, int NPCS = F16_PIECES>
struct Rs16ImageLds {
  typedef Rs16Geom<CIN, NPCS> G;
  static constexpr int NU = G::KS * G::NCH * 4 * G::NO, NB = 7 / CIN + 2;
  static constexpr int BYTES = G::REC_BYTES + (((CIN + 1) & ~1) + NU) * 8 + (NU * NB + 2 * CIN + 8 + G::NRC * G::NO * 4) * 4 + ((CIN * 2 + 15) & ~15) + 2 * CIN + 64 + 64;
};

template <int CIN, int NPCS = F16_PIECES>
__global__ __launch_bounds__(CONV_THREADS) void conv1_image_kernel(const Conv1ImageArgsN batch) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  conv1_image_body<CIN, NPCS>(batch.a[blockIdx.x], lds_raw);
}

// ---- the forward kernel
typedef unsigned rs16_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned rs16_u32x2 __attribute__((ext_vector_type(2)));

// the staged row segment of a wave's 32-pixel strip: pixels x0 - 2 .. x0 + 33 as they lie in the image
template <int CIN>
struct Rs16Seg {
  typedef Rs16Geom<CIN> G;
  static constexpr int P = G::P, RK = G::RK, NCH = G::NCH;
  static constexpr bool ODD = (CIN & 1) != 0;
  static constexpr int NCOPY = ODD ? 2 : 1;                                 // (odd channel counts: a second copy displaced by one half)
  static constexpr int PIX_BYTES = (32 + 2 * P) * CIN * 2;                  // the 36 pixels
  static constexpr int NREAL = (PIX_BYTES + 4 + (ODD ? 2 : 0) + 15) / 16;   // 16-byte pieces loaded from the image (18 channels: the + 4 are the last window's ones slots)
  static constexpr int WIN_BYTES = (31 * CIN + RK * (NCH - 1) + 32 + (ODD ? 1 : 0)) * 2;      // what the last pixel's last operand window reaches
  static constexpr int NPIECE = (WIN_BYTES + 15) / 16 > NREAL ? (WIN_BYTES + 15) / 16 : NREAL; // (pieces behind the loaded ones are zeros: slots whose weights are zero)
  static constexpr int CPYB = NPIECE * 16;
  static constexpr int SEGB = NCOPY * CPYB + 16;                            // (+ a dump slot for the lanes a store does not concern)
  static constexpr int NLOAD = (NPIECE + 63) / 64;                          // loads per copy, row and lane
};

template <int CIN>
__global__ __launch_bounds__(CONV_THREADS, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_fwd_rs16_kernel(const ConvArgsN batch) {
  typedef Rs16Geom<CIN> G;
  typedef Rs16Seg<CIN> SG;
  constexpr int KS = G::KS, P = G::P, NCH = G::NCH, NPC = G::NPC, RK = G::RK, NSET = G::NSET;
  constexpr bool ODDC = SG::ODD;
  constexpr int NCOPY = SG::NCOPY, NLOAD = SG::NLOAD, CPYB = SG::CPYB, SEGB = SG::SEGB;
  static_assert(NCH >= 1 && NCH <= 3, "at most three chunks per row: 30 A operands = 120 registers");
  static_assert((NCH * NSET) % 2 == 0, "the two operand buffers alternate with the chunk count: a block of NSET rows must hold an even number");
  constexpr int W = 64, Wp = 32;
  const ConvArgs& a = batch.a[blockIdx.y];
  // LDS: the restart values; per wave the pooled row on its way to the stores (160 f32 + 160 code bytes + a dump slot) and two staged
  // input rows of its strip
  constexpr int TRB = 640 + 160 + 16;
  constexpr int WVB = TRB + 2 * SEGB;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[G::CT_BYTES + 4 * WVB];
  float* ct_lds = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned char* rec = reinterpret_cast<const unsigned char*>(a.wimg);
  for (int i = tid; i < G::CT_BYTES / 16; i += CONV_THREADS)
    reinterpret_cast<f32x4*>(ct_lds)[i] = reinterpret_cast<const f32x4*>(rec + G::CT_OFF)[i];
  __syncthreads();
  const int li = lane & 15, lj = lane >> 4;
  const int swave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sstrip = swave & 1, simg = swave >> 1;
  // Two bands of output rows per image (a.nbands == 2: the launcher's answer to a grid that would leave every SIMD with ONE wave -- NAF's two
  // trunks, a single network's forward): band 0 computes output rows [0, band_rows), band 1 the rest; a band's walk starts two input rows
  // above its first output row, on the unrolled loop's period (band_rows - 2 is a multiple of NSET), and the output rows its first steps
  // complete with sums that miss the rows above are never stored.  Every stored output adds the same products in the same order as without bands.
  const int nbands = a.nbands > 1 ? 2 : 1;
  const int band = nbands > 1 ? ((int)blockIdx.x & 1) : 0;
  const int sb = ((int)blockIdx.x / nbands) * 2 + simg;
  if (sb >= a.B) return;
  const int H = a.H, Hp = H >> 1, nout = a.nout;
  const int y_lo = band ? a.band_rows : 0, y_hi = (nbands > 1 && band == 0) ? a.band_rows : H;     // the band's output rows
  const int q_b = band ? y_lo - P : 0;                         // first input row of its walk

  // ---- weights: KS x NCH x NPC A operands, resident
  f16x8 wv[KS][NCH][NPC];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int pc = 0; pc < NPC; ++pc)
        wv[ky][ch][pc] = *reinterpret_cast<const f16x8*>(rec + ((ky * NCH + ch) * NPC + pc) * 1024 + lane * 16);
  const float inv = *reinterpret_cast<const float*>(rec + G::SC_OFF);
  const unsigned* pwt = reinterpret_cast<const unsigned*>(rec + G::PW_OFF);
  const unsigned ones = pwt[15];                              // (chunk 0, lane group 3, dword 3: both ones slots)
  // the pivot of channel c sits where k = c does: pw[chunk 0][c >> 3][(c & 7) >> 1], half c & 1
  const auto pivot_word = [&](int c) __attribute__((always_inline)) { return pwt[(c >> 3) * 4 + ((c & 7) >> 1)]; };

  // ---- the image rows.  The strip's segment of a row is loaded as it lies in memory (16 contiguous bytes per lane: coalesced -- a lane
  // loading its own operand windows, as conv_k16.h does, asks the vector cache for 6 KB per row of which 1.3 KB are distinct, and that
  // cache was the bound), goes through registers into LDS two rows ahead of its use, the two pixels outside the image (SAME padding)
  // are overwritten with the pivots there, and the operand windows are LDS reads at per-lane addresses.
  const int rowbytes = W * CIN * 2;
  void* const in_base = (void*)((const char*)a.in + ((long)(a.img_slot ? a.img_slot[sb] : sb) * a.in_bstride) * 2 - 128);
  const int in_records = H * rowbytes + 128 + 256;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(in_base, 0, in_records, 0x00020000);
  unsigned char* xring = lds_raw + G::CT_BYTES + 4 * TRB + swave * 2 * SEGB;
  int gv[NCOPY][NLOAD];
  uint32_t xw[NCOPY][NLOAD];
#pragma unroll
  for (int cp = 0; cp < NCOPY; ++cp)
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
      const int piece = 64 * j + lane;
      // (pieces behind the loaded ones: an offset beyond the descriptor's range reads zeros; lanes without a piece store to the dump slot)
      gv[cp][j] = piece < SG::NREAL ? 128 + (sstrip * 32 - P) * CIN * 2 + 16 * piece - 2 * cp : 0x7FFFFFF0;
      xw[cp][j] = keep_in_vgpr(lds_addr(xring + (piece < SG::NPIECE ? cp * CPYB + 16 * piece : SEGB - 16)));
    }
  // the border pixels' pivots: an even channel count writes dword t of the two pixels (channels 2 t mod CIN, + 1); an odd one the
  // 2 CIN halves one by one, into both copies, and zeros over the halves between the 36th pixel and the end of the piece that holds it
  // (image bytes that zero weights multiply: they must be finite, and behind the last slot of a store lies a guard band nobody wrote)
  const int bhalf = sstrip == 0 ? 0 : (32 + P) * CIN;       // first half of the strip's two border pixels within the segment
  uint32_t xpw[NCOPY];
  unsigned pivw;
  if (!ODDC) {
    const int pt = lane % ((CIN / 2) > 0 ? (CIN / 2) : 1);
    pivw = pivot_word(2 * pt);
    xpw[0] = keep_in_vgpr(lds_addr(xring + (lane < CIN ? bhalf * 2 + 4 * lane : SEGB - 16)));
  } else {
    const int c = lane % CIN;
    pivw = lane < 2 * CIN ? (pivot_word(c) >> (16 * (c & 1))) & 0xFFFFu : 0u;
#pragma unroll
    for (int cp = 0; cp < NCOPY; ++cp) {
      const int nz = cp == 0 ? 4 : 3;                        // halves between the pixels' end and the piece's
      const int h = lane < 2 * CIN ? bhalf + lane : (32 + 2 * P) * CIN + (lane - 2 * CIN);
      xpw[cp] = keep_in_vgpr(lds_addr(xring + (lane < 2 * CIN + nz ? cp * CPYB + (h + cp) * 2 : SEGB - 16)));
    }
  }
  // operand windows: pixel 2 li + m of the strip, chunk ch, lane group lj: halves (2 li + m) CIN + RK ch + 8 lj .. + 7 of the segment
  const uint32_t xrd = keep_in_vgpr(lds_addr(xring + (2 * li * CIN + 8 * lj) * 2));
  const uint32_t xrd1 = keep_in_vgpr(lds_addr(xring + SEGB + (2 * li * CIN + 8 * lj) * 2));
  rs16_u32x4 stage[NCOPY][NLOAD];
  auto load_row = [&](int q) __attribute__((always_inline)) {
#pragma unroll
    for (int cp = 0; cp < NCOPY; ++cp)
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) stage[cp][j] = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, gv[cp][j], q * rowbytes, 0);
  };
  auto stage_row = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int cp = 0; cp < NCOPY; ++cp)
#pragma unroll
      for (int j = 0; j < NLOAD; ++j) lds_store(xw[cp][j], slot * SEGB, stage[cp][j]);
    if (!ODDC) lds_store(xpw[0], slot * SEGB, pivw);
    else {
#pragma unroll
      for (int cp = 0; cp < NCOPY; ++cp) lds_store_u16(xpw[cp], slot * SEGB, pivw);
    }
  };
  k16_u32x4 xb[2][2];
  auto read_x = [&](int buf, int ch, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      // (the odd pixels of an odd channel count: the displaced copy, where they start on a dword)
      const int off = (ODDC && m == 1) ? CPYB + (CIN + 1 + RK * ch) * 2 : (m * CIN + RK * ch) * 2;
      unsigned t[4];                                           // (dword reads, which the compiler pairs into ds_read2_b32)
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] = lds_load<unsigned>(slot ? xrd1 : xrd, off + 4 * i);
      xb[buf][m] = (k16_u32x4){t[0], t[1], t[2], lj == 3 ? ones : t[3]};
    }
  };

  // ---- outputs.  The pooled row of the wave's strip is 16 pixels x nout values, contiguous in NHWC: lane l < 4 nout writes elements
  // 4 l .. 4 l + 3 of it (16 / 8 / 4 contiguous bytes per lane: f32, bf16 planes, codes)
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out ? a.out + (long)sb * a.out_bstride : (float*)a.w, 0, a.out ? Hp * Wp * nout * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t b16_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_b16 ? a.out_b16 + (long)sb * (Hp * Wp * nout) : (unsigned short*)a.w, 0,
      a.out_b16 ? (int)(2 * a.out_b16_plane * 2 + (long)Hp * Wp * nout * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t amax_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      a.out_amax ? a.out_amax + (long)sb * Hp * Wp * nout : (uint8_t*)a.w, 0, a.out_amax ? Hp * Wp * nout : 0, 0x00020000);
  const int plane_bytes = (int)(a.out_b16_plane * 2);
  constexpr unsigned BIG = 0x08000000u;                      // beyond every descriptor's range, also x 2 and x 4
  const unsigned eL = (lane < 4 * nout) ? (unsigned)(sstrip * 16 * nout + 4 * lane) : BIG;      // element offset of the lane's four values within a pooled image row
  unsigned char* trw = lds_raw + G::CT_BYTES + swave * TRB;
  // (pooled layout -> row layout through LDS: a lane holds filters 4 lj .. 4 lj + 3 of pooled pixel li; invalid filters go to the dump slot)
  const uint32_t twA = keep_in_vgpr(lds_addr(trw + ((4 * lj + 1 < nout) ? (li * nout + 4 * lj) * 4 : 800)));
  const uint32_t twB = keep_in_vgpr(lds_addr(trw + ((4 * lj + 3 < nout) ? (li * nout + 4 * lj + 2) * 4 : 808)));
  const uint32_t tcA = keep_in_vgpr(lds_addr(trw + ((4 * lj + 1 < nout) ? 640 + li * nout + 4 * lj : 800)));
  const uint32_t tcB = keep_in_vgpr(lds_addr(trw + ((4 * lj + 3 < nout) ? 640 + li * nout + 4 * lj + 2 : 808)));
  const uint32_t trd = keep_in_vgpr(lds_addr(trw + (lane < 4 * nout ? 16 * lane : 0)));
  const uint32_t trc = keep_in_vgpr(lds_addr(trw + 640 + (lane < 4 * nout ? 4 * lane : 0)));

  // restart values of this lane's accumulator elements: ct[rc][x class of its column in tile m][4 lj .. 4 lj + 3]
  uint32_t ctadr[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int xs = (sstrip == 0 && li == 0) ? 1 + m : ((sstrip == 1 && li == 15) ? 3 + m : 0);
    ctadr[m] = keep_in_vgpr(lds_addr(ct_lds + xs * 16 + 4 * lj));
  }
  auto ct_load = [&](int rc, int m) __attribute__((always_inline)) { return lds_load<f32x4>(ctadr[m] + (uint32_t)(rc * 5 * 16 * 4), 0); };
  const auto row_class = [&](int y) __attribute__((always_inline)) { return y < P ? y : (y >= H - P ? 2 * P - (H - 1 - y) : P); };

  f32x4 acc[NSET][2];
  f32x4 cin[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    cin[m] = ct_load(P, m);
    acc[0][m] = ct_load(0, m); acc[1][m] = ct_load(1, m);
#pragma unroll
    for (int s = 2; s < NSET; ++s) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float tv[4] = {0.f, 0.f, 0.f, 0.f};
  bool xgt[4] = {false, false, false, false};
  load_row(q_b);
  stage_row(0);
  load_row(q_b + 1);
  read_x(0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);

  auto step = [&](auto sqtag, auto gentag, const int q) __attribute__((always_inline)) {
    constexpr int SQ = decltype(sqtag)::value;
    constexpr bool GEN = decltype(gentag)::value;
    constexpr int SD = (SQ + 3) % NSET;                     // the set of output row q - 3: complete since the last step
    constexpr bool ODD = (SQ & 1) == 0;                      // q - 3 is odd: a pooled row is complete
    constexpr int SLOT = SQ & 1;                             // the LDS slot of row q (q0 is a multiple of NSET = 6)
    const int yd = q - 3;
    f32x4 cg[2] = {cin[0], cin[1]};                          // restart value of the set output row q + 2 starts in
    if (GEN) {
      const int rc = row_class(q + 2 < H ? q + 2 : P);
      cg[0] = ct_load(rc, 0); cg[1] = ct_load(rc, 1);
    }
    // ---- epilogue of row yd: an even row's x-pooled values wait in tv / xgt; an odd row completes a pooled row, which goes through
    // the wave's LDS slot into row layout
    float pv[4]; int code[4];
    auto pool = [&](const int r0, const int r1) __attribute__((always_inline)) {      // filters r0 .. r1 - 1 of the lane's four
      if (ODD) {
#pragma unroll
        for (int r = r0; r < r1; ++r) {
          const float z0 = acc[SD][0][r], z1 = acc[SD][1][r];
          const bool xgb = z1 > z0;
          const float bv = xgb ? z1 : z0;
          const bool lower = bv > tv[r];
          const float mx = lower ? bv : tv[r];
          const bool act = mx > 0.f;
          code[r] = (lower ? (xgb ? 3 : 2) : (xgt[r] ? 1 : 0)) | (act ? POOL_ACTIVE : 0);
          pv[r] = act ? mx * inv : 0.f;
        }
        if (r0 == 0) {
          lds_store(twA, 0, (f32x2){pv[0], pv[1]});
          lds_store_u16(tcA, 0, (unsigned)(code[0] | (code[1] << 8)));
        } else {
          lds_store(twB, 0, (f32x2){pv[2], pv[3]});
          lds_store_u16(tcB, 0, (unsigned)(code[2] | (code[3] << 8)));
        }
      } else {
#pragma unroll
        for (int r = r0; r < r1; ++r) {
          const float z0 = acc[SD][0][r], z1 = acc[SD][1][r];
          xgt[r] = z1 > z0;
          tv[r] = xgt[r] ? z1 : z0;
        }
      }
    };
    auto stores = [&]() __attribute__((always_inline)) {
      const f32x4 pr = lds_load<f32x4>(trd, 0);
      const unsigned cd = lds_load<unsigned>(trc, 0);
      const bool live = !GEN || (yd >= y_lo && yd < y_hi);
      const unsigned eo = live ? eL : BIG;
      const int orow = (yd >> 1) * Wp * nout;
      unsigned hb[4], mb[4], lb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hb[r] = __float_as_uint(pr[r]) & 0xFFFF0000u;
        const float r1 = pr[r] - __uint_as_float(hb[r]);
        mb[r] = __float_as_uint(r1) & 0xFFFF0000u;
        lb[r] = __float_as_uint(r1 - __uint_as_float(mb[r]));
      }
      buffer_store_b128_held((rs16_u32x4){__float_as_uint(pr[0]), __float_as_uint(pr[1]), __float_as_uint(pr[2]), __float_as_uint(pr[3])}, out_rsrc, (int)(eo * 4), orow * 4);
      __builtin_amdgcn_raw_buffer_store_b64((rs16_u32x2){(hb[0] >> 16) | hb[1], (hb[2] >> 16) | hb[3]}, b16_rsrc, (int)(eo * 2), orow * 2, 0);
      __builtin_amdgcn_raw_buffer_store_b64((rs16_u32x2){(mb[0] >> 16) | mb[1], (mb[2] >> 16) | mb[3]}, b16_rsrc, (int)(eo * 2), plane_bytes + orow * 2, 0);
      __builtin_amdgcn_raw_buffer_store_b64((rs16_u32x2){(lb[0] >> 16) | (lb[1] & 0xFFFF0000u), (lb[2] >> 16) | (lb[3] & 0xFFFF0000u)}, b16_rsrc, (int)(eo * 2),
                                            2 * plane_bytes + orow * 2, 0);
      __builtin_amdgcn_raw_buffer_store_b32(cd, amax_rsrc, (int)eo, orow, 0);
    };
    auto mfmas = [&](auto chtag) __attribute__((always_inline)) {
      constexpr int ch = decltype(chtag)::value;
      constexpr int BUF = (NCH * SQ + ch) & 1;
#pragma unroll
      for (int pc = NPC - 1; pc >= 0; --pc)
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
          const int s = (SQ + 2 - ky + NSET) % NSET;         // output row q + 2 - ky
          const bool first = ky == 0 && ch == 0 && pc == NPC - 1;
          acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ky][ch][pc], __builtin_bit_cast(f16x8, xb[BUF][0]), first ? cg[0] : acc[s][0], 0, 0, 0);
          acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ky][ch][pc], __builtin_bit_cast(f16x8, xb[BUF][1]), first ? cg[1] : acc[s][1], 0, 0, 0);
        }
    };
    // chunk ch of row q: the NEXT chunk's operands are requested (the next row's first chunk behind the last one), then this chunk's
    // MFMAs, then a share of row yd's epilogue
    auto chunk = [&](auto chtag) __attribute__((always_inline)) {
      constexpr int ch = decltype(chtag)::value;
      constexpr int NB = (NCH * SQ + ch + 1) & 1;
      if (ch + 1 < NCH) read_x(NB, ch + 1, SLOT); else read_x(NB, 0, SLOT ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mfmas(chtag);
      if (ch == 0) pool(0, 2);
      if (ch == (NCH > 1 ? 1 : 0)) pool(2, 4);
      if (ch == NCH - 1 && ODD) stores();
    };
    if (!GEN || q < H) {
      // row q + 1 (in the staging registers since the last step) -> LDS; row q + 2 -> staging registers
      stage_row(SLOT ^ 1);
      load_row(q + 2);                                       // (behind the last row: beyond the descriptor's range, zeros nobody uses)
      chunk(std::integral_constant<int, 0>{});
      if (NCH > 1) chunk(std::integral_constant<int, (NCH > 1 ? 1 : 0)>{});
      if (NCH > 2) chunk(std::integral_constant<int, (NCH > 2 ? 2 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
    } else { pool(0, 2); pool(2, 4); if (ODD) stores(); }    // (steps behind the image: the last pooled rows)
  };
  auto block = [&](auto gentag, const int q0) __attribute__((always_inline)) {
    constexpr bool GEN = decltype(gentag)::value;
    {  // issue arbitration is by priority, then age: the older of a SIMD's two waves ran ahead (64 vs 88 us of a 91 us launch) and the
       // younger one finished alone at half the pipe's rate; the priorities rotate every NSET rows (as conv_k16.h's do)
      const int pr = ((int)blockIdx.y + q0 / NSET) & 3;
      if (pr == 0) __builtin_amdgcn_s_setprio(0); else if (pr == 1) __builtin_amdgcn_s_setprio(1);
      else if (pr == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3);
    }
    if (GEN && q0 + 0 >= y_hi + 3) return; step(std::integral_constant<int, 0>{}, gentag, q0 + 0);
    if (GEN && q0 + 1 >= y_hi + 3) return; step(std::integral_constant<int, 1>{}, gentag, q0 + 1);
    if (GEN && q0 + 2 >= y_hi + 3) return; step(std::integral_constant<int, 2>{}, gentag, q0 + 2);
    if (GEN && q0 + 3 >= y_hi + 3) return; step(std::integral_constant<int, 3>{}, gentag, q0 + 3);
    if (GEN && q0 + 4 >= y_hi + 3) return; step(std::integral_constant<int, 4>{}, gentag, q0 + 4);
    if (GEN && q0 + 5 >= y_hi + 3) return; step(std::integral_constant<int, 5>{}, gentag, q0 + 5);
  };
  // the first block is a general one (the image's top rows / a band's unstored first rows: q_b + NSET - 1 - 3 = y_lo + 1, so every later
  // step stores a row of the band); interior blocks while every step's output row is stored and every input row it asks for is interior
  int q0 = q_b;
  block(std::true_type{}, q0); q0 += NSET;
  const int q_int = min(H - 4, y_hi + 3);
  for (; q0 + NSET <= q_int; q0 += NSET) block(std::false_type{}, q0);
  for (; q0 < y_hi + 3; q0 += NSET) block(std::true_type{}, q0);
}

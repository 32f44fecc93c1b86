// conv1 forward on the f16 matrix pipes (conv_k16.h) for TWO networks that read the SAME images: the actor and the critic both
// take state_1, the two target networks state_2 (ddpg_cartpole.py:333-334, :199) with the same whitening tables.  One workgroup
// computes both networks' outputs of its images:
//
//   * the column space of the (ky,o) formulation becomes N = (network, p, o) = 2 x 50 = 100 of 112 columns = SEVEN 16-column tiles
//     instead of 2 x 4 (50 of 64 each): 12.5 % fewer MFMAs for the same outputs;
//   * the A operands (raw pixels, straight from global memory), their masks and the ones-channel constants are loaded and prepared
//     ONCE per row for both networks: per row a wave issues 126 MFMAs where two one-network waves issue 144, next to the same 6
//     operand loads and 20 mask operations instead of twice that;
//   * both weight images sit in LDS (2 x 50.7 KB at 5x5x18): one workgroup per CU, one wave per SIMD -- the accumulators (7 tiles x
//     2 M tiles) and a full B register set (7 x 3 pieces) fit the 512 registers a lone wave has.
//
// A column j = 16 t + li of tile t belongs to network j / 50; everything per column (weight address, bias, pool-pair buffer) is
// per-lane state exactly as in conv_k16.h, so the row loop is the one-network loop with NT = 7.  The epilogue extracts, per
// completed output row, the ten columns of block p of EACH network, and the pooled-row writer runs once per network.
// Restrictions (the launcher checks them): pooled output (no batch norm), f16 images with even CIN, no conv3 tail.
#pragma once
#include "conv_k16.h"

template <int CIN, int KS, int XT, int IPW>
struct K16PairGeom {
  typedef K16Geom<CIN, KS, XT, IPW, false> G1;
  static constexpr int NO = KYO_NO, P = KS / 2, NNET = 2;
  static constexpr int NCOL = NNET * KS * NO;               // 100
  static constexpr int NT = (NCOL + 15) / 16;               // 7
  static constexpr int WLB = G1::WLB, EF = G1::EF;
  static constexpr int LDS_BYTES = NNET * WLB + NNET * 4 * EF * 4 + 64;
  static_assert((CIN & 1) == 0, "even channel counts (4-byte aligned operand windows)");
};

template <int CIN, int KS, int XT, int IPW>
__global__ __launch_bounds__(CONV_THREADS, 1) void conv_fwd_k16_pair_kernel(const ConvArgsN batch) {
  typedef K16Geom<CIN, KS, XT, IPW, false> G;
  typedef K16PairGeom<CIN, KS, XT, IPW> GP;
  constexpr int P = G::P, NT = GP::NT, NCH = G::NCH, NPC = G::NPC, NO = KYO_NO, NNET = 2, NCB = KS * NO;      // NCB: columns of one network
  const ConvArgs& a = batch.a[2 * blockIdx.y];        // geometry, images, whitening: shared by the pair
  const ConvArgs& a1 = batch.a[2 * blockIdx.y + 1];   // (references into the kernel arguments, selected with compile-time k: a pointer array
#define K16P_NET(k) ((k) == 0 ? a : a1)               //  indexed per lane would put the whole argument block into scratch memory)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* wl = lds_raw;                                              // [NNET] weight images
  float2* ebuf = reinterpret_cast<float2*>(lds_raw + NNET * G::WLB);        // [NNET][4 waves][2 parities][8*XT][NO]
  float* red = reinterpret_cast<float*>(lds_raw + NNET * G::WLB + NNET * 4 * G::EF * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  const int strip = wave % G::STRIPS;
  const int nbands = a.nbands > 1 ? a.nbands : 1;
  const int band = (int)blockIdx.x % nbands;
  const int b0 = ((int)blockIdx.x / nbands) * IPW;
  const int H = a.H, W = a.W, Hp = H >> 1, Wp = W >> 1, nout = a.nout;
  const int ymin = band * a.band_rows;
  const int qbeg = band > 0 ? ymin - P : 0;
  const int qend = (nbands > 1 && band + 1 < nbands ? ymin + a.band_rows : H) + P;

  // column j = 16 t + li -> (network, p, o)
  int cnet[NT], cp[NT], co[NT]; bool cvalid[NT];
  float braw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = 16 * t + li;
    cvalid[t] = j < GP::NCOL;
    cnet[t] = cvalid[t] ? j / NCB : 0;
    const int jj = j - cnet[t] * NCB;
    cp[t] = cvalid[t] ? jj / NO : 0;
    co[t] = jj % NO;
    const bool bok = cvalid[t] && co[t] < nout;
    const float bz0 = a.bias[bok ? co[t] : 0], bz1 = a1.bias[bok ? co[t] : 0];
    braw[t] = bok ? (cnet[t] ? bz1 : bz0) : 0.f;
  }

  // ---- one-time setup: both split weight images (conv_k16.h, per network)
  float sc[NNET], inv[NNET];
  {
    constexpr int NV = KS * NCH * 32 * NO;            // (ky, k, o), o fastest
    constexpr int NW = (NV + CONV_THREADS - 1) / CONV_THREADS;
    float wv[NNET][NW];
    float vmax[NNET] = {0.f, 0.f};
    float* onesw = reinterpret_cast<float*>(ebuf);    // [NNET][KS][KS][NO] scratch (the pool-pair buffers are not in use yet)
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        const int i = tid + n * CONV_THREADS;
        const int o = i % NO, r = i / NO;
        const int kk = r % (NCH * 32), ky = r / (NCH * 32);
        const bool real = i < NV && o < nout && kk < G::KROW;
        const float w = K16P_NET(k).w[real ? (ky * G::KROW + kk) * nout + o : 0], sck = a.scale[real ? kk % CIN : 0];
        wv[k][n] = real ? w * sck : 0.f;
      }
      const int o = tid % NO, kq = tid / NO;          // kq = ky * KS + kx
      const bool act = tid < KS * KS * NO && o < nout;
      float wq[CIN], sh[CIN];
#pragma unroll
      for (int c = 0; c < CIN; ++c) { wq[c] = K16P_NET(k).w[act ? (kq * CIN + c) * nout + o : 0]; sh[c] = a.shift[c]; }
      float sacc = 0.f;
#pragma unroll
      for (int c = 0; c < CIN; ++c) sacc += wq[c] * sh[c];
      if (tid < KS * KS * NO) onesw[k * KS * KS * NO + tid] = act ? sacc : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        const int i = tid + n * CONV_THREADS;
        const int o = i % NO, r = i / NO;
        const int kk = r % (NCH * 32), ky = r / (NCH * 32);
        float v = wv[k][n];
        if (i < NV && kk >= G::KROW && kk < G::KAUG) v = onesw[k * KS * KS * NO + (ky * KS + (kk - G::KROW)) * NO + o];      // ones channel: sum_c W t_c
        if (K16P_NET(k).wscale != 0.f) v *= K16P_NET(k).wscale;
        wv[k][n] = v;
        vmax[k] = fmaxf(vmax[k], fabsf(v));
      }
      for (int o = 32; o > 0; o >>= 1) vmax[k] = fmaxf(vmax[k], __shfl_xor(vmax[k], o));
      if (lane == 0) red[4 * k + wave] = vmax[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NNET; ++k) {
      const float vm = fmaxf(fmaxf(red[4 * k], red[4 * k + 1]), fmaxf(red[4 * k + 2], red[4 * k + 3]));
      int S = 0;
      if (vm > 0.f && vm < 3.0e38f) S = 14 - ilogbf(vm);       // vm 2^S in [2^14, 2^15)
      S = S > 100 ? 100 : (S < -100 ? -100 : S);
      sc[k] = ldexpf(1.f, S); inv[k] = ldexpf(1.f, -S);
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        const int i = tid + n * CONV_THREADS;
        if (i < NV) {
          const int o = i % NO, r = i / NO;
          const int kk = r % (NCH * 32), ky = r / (NCH * 32);
          const int ch = kk >> 5, g = (kk >> 3) & 3, e = kk & 7;
          const float v = wv[k][n] * sc[k];
          const _Float16 hh = (_Float16)v;
          const float r1 = v - (float)hh;
          const _Float16 mm = (_Float16)r1;
          const float r2 = r1 - (float)mm;
          const _Float16 ll = (_Float16)r2;
          unsigned char* dst = wl + k * G::WLB + ky * G::PS + g * G::GS + o * 16 + e * 2;
          *reinterpret_cast<unsigned short*>(dst + (ch * NPC + 0) * G::SLAB) = __builtin_bit_cast(unsigned short, hh);
          *reinterpret_cast<unsigned short*>(dst + (ch * NPC + 1) * G::SLAB) = __builtin_bit_cast(unsigned short, mm);
          if (NPC > 2) *reinterpret_cast<unsigned short*>(dst + (ch * NPC + 2) * G::SLAB) = __builtin_bit_cast(unsigned short, ll);
        }
      }
    }
  }
  __syncthreads();                                   // weight images visible (and the scratch in ebuf dead); no barrier after this one

  const int swave = __builtin_amdgcn_readfirstlane(wave);
  const int simg = swave / G::STRIPS, sstrip = swave % G::STRIPS;
  const int sbimg = b0 + simg;
  if (sbimg >= a.B) return;                          // wave-uniform

  // ---- per-lane column bookkeeping: weight address by row phase, pool-pair address, scaled bias
  uint32_t wadr[KS][NT];
  uint32_t eadr[NT];
  float biast[NT];
  // pool-pair buffers: [net][wave] blocks of 2 * 8 * XT * NO float2
  float2* ev0 = ebuf + swave * (2 * 8 * XT * NO);
  constexpr int EVNET = 4 * (2 * 8 * XT * NO);        // float2 between the two networks' buffers
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    biast[t] = braw[t] * (cnet[t] ? sc[1] : sc[0]);
    eadr[t] = keep_in_vgpr(lds_addr(ev0 + cnet[t] * EVNET + (lj * 2) * NO + co[t]));
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int ky = (sq + P - cp[t] + KS) % KS;
      wadr[sq][t] = keep_in_vgpr(lds_addr(wl + cnet[t] * G::WLB + ky * G::PS + lj * G::GS + (cvalid[t] ? co[t] : 0) * 16));
    }
  }

  // ---- operand masks (conv_k16.h): shared by both networks
  unsigned emask[NCH][XT][4], ecst[NCH][XT][4];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int m = 0; m < XT; ++m) {
        unsigned cs = 0u, bm = 0u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = 32 * ch + 8 * lj + 2 * v + h;
          const int kx = k < G::KROW ? k / CIN : k - G::KROW;
          const int xi = strip * G::SW + m * 16 + li + kx - P;
          const bool inside = xi >= 0 && xi < W;
          if (k < G::KROW && inside) bm |= 0xFFFFu << (16 * h);
          if (k >= G::KROW && k < G::KAUG && inside) cs |= 0x3C00u << (16 * h);      // f16 1.0
        }
        ecst[ch][m][v] = cs;
        emask[ch][m][v] = bm;
      }

  // ---- pooled-row writer (per network: same lanes, same offsets, its own descriptors)
  constexpr int NC = (8 * XT * (NO / 2) + 63) / 64;
  uint32_t cadr[NC];
  unsigned coe[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int idx = lane + 64 * i;
    const int hn = nout >> 1;
    const int xl = idx / hn, o = 2 * (idx - xl * hn);
    const int px = ((sstrip * G::SW) >> 1) + xl;
    const bool act = idx < 8 * XT * hn && px < Wp;
    cadr[i] = keep_in_vgpr(lds_addr(ev0 + (act ? xl * NO + o : 0)));
    coe[i] = act ? (unsigned)(px * nout + o) : 0x3FFFFFFCu;      // inactive lanes: beyond every descriptor's range (x 1, 2, 4)
  }
  constexpr int ST = NNET * NC * 5;                   // vector-memory stores of one writer pass (all issued)
  __amdgpu_buffer_rsrc_t b16_rsrc[NNET], out_rsrc[NNET], amax_rsrc[NNET];
  int b16_plane_bytes[NNET];
#pragma unroll
  for (int k = 0; k < NNET; ++k) {
    const ConvArgs& ak = K16P_NET(k);
    b16_rsrc[k] = __builtin_amdgcn_make_buffer_rsrc(
        ak.out_b16 ? ak.out_b16 + (long)sbimg * (Hp * Wp * nout) : (unsigned short*)ak.w, 0,
        ak.out_b16 ? (int)(2 * ak.out_b16_plane * 2 + (long)Hp * Wp * nout * 2) : 0, 0x00020000);
    b16_plane_bytes[k] = (int)(ak.out_b16_plane * 2);
    out_rsrc[k] = __builtin_amdgcn_make_buffer_rsrc(ak.out ? ak.out + (long)sbimg * ak.out_bstride : (float*)ak.w, 0, ak.out ? Hp * Wp * nout * 4 : 0, 0x00020000);
    amax_rsrc[k] = __builtin_amdgcn_make_buffer_rsrc(ak.out_amax ? ak.out_amax + (long)sbimg * Hp * Wp * nout : (uint8_t*)ak.w, 0, ak.out_amax ? Hp * Wp * nout : 0, 0x00020000);
  }

  // ---- A operands: 16 bytes per lane and (tile, chunk) straight from the image row, once for both networks
  const int rowbytes = W * CIN * 2;
  void* const in_base = (void*)((const char*)a.in + ((long)(a.img_slot ? a.img_slot[sbimg] : sbimg) * a.in_bstride) * 2 - G::BIAS_BYTES);
  const int in_records = H * rowbytes + G::BIAS_BYTES + 256;
  const k16_i32x4 in_desc0 = k16_raw_desc(in_base, in_records);
  // (wave-uniform by construction; said explicitly so that the "s" operands of the inline-asm loads are scalar registers)
  const k16_i32x4 in_desc = {__builtin_amdgcn_readfirstlane(in_desc0.x), __builtin_amdgcn_readfirstlane(in_desc0.y),
                             __builtin_amdgcn_readfirstlane(in_desc0.z), __builtin_amdgcn_readfirstlane(in_desc0.w)};
  const int avoff = G::BIAS_BYTES + ((strip * G::SW + li - P) * CIN + 8 * lj) * 2;
  k16_u32x4 av[NCH][XT];
  constexpr int LPC = XT;                             // vector-memory loads per chunk and row
  auto load_a = [&](int ch, int y) {
#pragma unroll
    for (int m = 0; m < XT; ++m)
      k16_issue_b128(av[ch][m], in_desc, avoff, __builtin_amdgcn_readfirstlane(y * rowbytes + (m * 16 * CIN + 32 * ch) * 2));
  };

  f32x4 acc[XT][NT];
#pragma unroll
  for (int m = 0; m < XT; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) load_a(ch, qbeg);
  const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0, 0x00020000);      // empty range: stores are dropped
#pragma unroll
  for (int i = 0; i < ST; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, null_rsrc, 64 * i, 0, 0);      // row 0 sees the same sequence as every other row

  // B operands of ONE k chunk, all seven tiles; rolling reload piece by piece (conv_k16.h)
  f16x8 bv[NT][NPC];
  auto load_b_piece = [&](int ch, int pc, const uint32_t* wa) {
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t][pc] = lds_load<f16x8>(wa[t], (ch * NPC + pc) * G::SLAB);
  };
#pragma unroll
  for (int pc = 0; pc < NPC; ++pc) load_b_piece(0, pc, wadr[0]);

  for (int q0 = qbeg; q0 < qend; q0 += KS) {
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int q = q0 + sq;
      if (q >= qend) break;                          // uniform
      constexpr int PD_BASE = (KS - P) % KS;
      const int pdone = (PD_BASE + sq) % KS;
      if (q < H) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          k16_wait_vm<(NCH - 1) * LPC + ST>(av[ch][0]);
#pragma unroll
          for (int m = 0; m < XT; ++m) asm volatile("" : "+v"(av[ch][m]));
          k16_u32x4 af[XT];
#pragma unroll
          for (int m = 0; m < XT; ++m) {
            k16_u32x4 u = av[ch][m];
#pragma unroll
            for (int v = 0; v < 4; ++v)
              if (G::vgpr_may_need_mask(ch, m, v)) u[v] = (u[v] & emask[ch][m][v]) | ecst[ch][m][v];
            af[m] = u;
          }
#pragma unroll
          for (int pc = NPC - 1; pc >= 0; --pc) {    // small pieces first; XT*NT independent accumulators between the pieces
#pragma unroll
            for (int m = 0; m < XT; ++m)
#pragma unroll
              for (int t = 0; t < NT; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[m]), bv[t][pc], acc[m][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ch + 1 < NCH) load_b_piece(ch + 1, pc, wadr[sq]);
            else if (q + 1 < H) load_b_piece(0, pc, wadr[(sq + 1) % KS]);
            __builtin_amdgcn_sched_barrier(0);
          }
          load_a(ch, q + 1);                         // this chunk's operands of the next row (behind the last row: masked by the descriptor)
        }
      }
      const int y = q - P >= ymin ? q - P : -1;
      const int par = y & 1;
      // extraction: the NO columns of block pdone of EACH network
#pragma unroll
      for (int k = 0; k < NNET; ++k)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int clo = k * NCB + pdone * NO - 16 * t, chi = clo + NO - 1;
          if (chi >= 0 && clo <= 15) {
            const bool inr = li >= clo && li <= chi;
            if (inr) {
#pragma unroll
              for (int m = 0; m < XT; ++m) {
                if (y >= 0) {
                  const uint32_t ea = eadr[t] + (uint32_t)(par * (8 * XT * NO) * 8);
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const float z0 = acc[m][t][2 * h], z1 = acc[m][t][2 * h + 1];
                    lds_store(ea, ((m * 8 + h) * NO) * 8, (f32x2){z1 > z0 ? z1 : z0, __int_as_float(z1 > z0 ? 1 : 0)});
                  }
                }
                acc[m][t] = (f32x4){biast[t], biast[t], biast[t], biast[t]};
              }
            }
          }
        }
      const bool writer_row = y >= 0 && par == 1 && (y >> 1) < Hp;
      if (!writer_row && q < H) {
#pragma unroll
        for (int i = 0; i < ST; ++i) __builtin_amdgcn_raw_buffer_store_b32(0u, null_rsrc, 64 * i, 0, 0);
      }
      if (writer_row) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int orow = (y >> 1) * Wp * nout;
#pragma unroll
        for (int k = 0; k < NNET; ++k)
#pragma unroll
          for (int i = 0; i < NC; ++i) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const f32x4 top = lds_load<f32x4>(cadr[i], k * EVNET * 8), bot = lds_load<f32x4>(cadr[i], k * EVNET * 8 + (8 * XT * NO) * 8);   // (value, code) x 2
            float pv[2]; int code[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const bool lower = bot[2 * e] > top[2 * e];
              const float mx = lower ? bot[2 * e] : top[2 * e];
              code[e] = lower ? 2 + __float_as_int(bot[2 * e + 1]) : __float_as_int(top[2 * e + 1]);
              pv[e] = mx > 0.f ? mx * inv[k] : 0.f;
            }
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(pv[0]), __float_as_uint(pv[1])}, out_rsrc[k], (int)(coe[i] * 4), orow * 4, 0);
            unsigned hb[2], mb[2], lb[2];             // the next layer's A operand: three bf16 planes (truncating split, exact)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              hb[e] = __float_as_uint(pv[e]) & 0xFFFF0000u;
              const float r1 = pv[e] - __uint_as_float(hb[e]);
              mb[e] = __float_as_uint(r1) & 0xFFFF0000u;
              lb[e] = __float_as_uint(r1 - __uint_as_float(mb[e]));
            }
            __builtin_amdgcn_raw_buffer_store_b32((hb[0] >> 16) | hb[1], b16_rsrc[k], (int)(coe[i] * 2), orow * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b32((mb[0] >> 16) | mb[1], b16_rsrc[k], (int)(coe[i] * 2), b16_plane_bytes[k] + orow * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b32((lb[0] >> 16) | (lb[1] & 0xFFFF0000u), b16_rsrc[k], (int)(coe[i] * 2), 2 * b16_plane_bytes[k] + orow * 2, 0);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(code[0] | (code[1] << 8)), amax_rsrc[k], (int)coe[i], orow, 0);
          }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the look-ahead loads behind the last row)
#undef K16P_NET
}

// networks 2j and 2j + 1 of the batch read the same images with the same whitening tables and want pooled outputs
static inline bool conv_fwd_k16_pairable(const ConvArgsN& batch) {
  if (batch.n < 2 || (batch.n & 1)) return false;
  for (int j = 0; j + 1 < batch.n; j += 2) {
    const ConvArgs &x = batch.a[j], &y = batch.a[j + 1];
    if (x.in != y.in || x.in_bstride != y.in_bstride || x.img_slot != y.img_slot || x.scale != y.scale || x.shift != y.shift ||
        x.n3_w || y.n3_w || x.white_bstride || y.white_bstride)
      return false;
  }
  return true;
}

template <int CIN, int KS, int XT, int IPW>
static inline int conv_fwd_k16_pair_launch_t(cpp_ctx* ctx, const ConvArgsN& batch) {
  typedef K16PairGeom<CIN, KS, XT, IPW> GP;
  const ConvArgs& a = batch.a[0];
  const size_t lds_bytes = (size_t)GP::LDS_BYTES;
  auto kern = conv_fwd_k16_pair_kernel<CIN, KS, XT, IPW>;
  static bool attr_done[CPP_MAX_DEVICES] = {};
  if (!attr_done[cpp_dev_slot(ctx)]) {
    HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done[cpp_dev_slot(ctx)] = true;
  }
  const int grid = ((a.B + IPW - 1) / IPW) * (a.nbands > 1 ? a.nbands : 1);
  hipLaunchKernelGGL(kern, dim3(grid, batch.n / 2), dim3(CONV_THREADS), lds_bytes, ctx->stream, batch);
  LAUNCH_CHECK();
  return 0;
}

// the instances (conv_fwd_k16_pair.hip); *handled stays false for a geometry without one
int conv_fwd_k16_pair_dispatch(cpp_ctx* ctx, int cin, int xt, int ipw, const ConvArgsN& a, bool* handled);

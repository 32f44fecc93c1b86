// TIMING PROBE (no parity: random operands, no whitening slots, no border handling) for a different conv1-forward formulation -- DESIGN.md 8:
// conv_fwd_k16_kernel's launch is its waves' non-MFMA instruction chain plus both co-resident waves' MFMA time.  Here a wave streams input
// rows like conv_k16.h (each row loaded once, 6 buffer_load_dwordx4 per lane), but N is the 10 filters padded to 16 and the KS = 5 output
// rows in flight are 5 ACCUMULATOR SETS (tile restarts through the first MFMA's C operand: no per-lane reset, no `inr` branches), the x half of
// the 2 x 2 pool is two accumulator elements of a lane and the y half the previous row's registers: no LDS transpose, no writer passes.
// 60 instead of 48 MFMAs per row (N = 16 of which 10 are filters, against 50 of 64).  Same grid as the shipped kernel at cfg3: 512 workgroups
// of 4 waves = 2 images x 2 strips of 32 pixels, 64 x 64 x 18 f16 images, two f16 pieces of the weights in LDS.
//   hipcc --offload-arch=gfx950 -O3 -o cartpoleplusplus_amd/lib/conv1_rowstream_probe profiles/diag/conv1_rowstream_probe.hip && cartpoleplusplus_amd/lib/conv1_rowstream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64, W = 64, CIN = 18, KS = 5, P = 2, NO = 10, NCH = 3, RK = 30, NPC = 2, XT = 2;
constexpr int SLAB = 1024;                          // one (ky, chunk, piece): [lj 4][n 16][8 halves]
constexpr int WLB = KS * NCH * NPC * SLAB;          // 30 KB

template <int STORES>                                // 0: no output stores, 1: b32 / b16 / b8 stores per value, 2: channel pairs through DPP
__global__ __launch_bounds__(256, 2) void probe(const _Float16* __restrict__ img, const _Float16* __restrict__ wimg, float* __restrict__ out,
                                                unsigned short* __restrict__ planes, unsigned char* __restrict__ codes, int nimg_per_set) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lj = lane >> 4;
  const int net = blockIdx.y;
  {  // the prebuilt weight image of this network: 30 KB, 16 bytes per thread and pass
    const u32x4* src = reinterpret_cast<const u32x4*>(wimg + (size_t)net * (WLB / 2));
    for (int i = tid; i < WLB / 16; i += 256) reinterpret_cast<u32x4*>(lds)[i] = src[i];
  }
  __syncthreads();
  const int simg = wave >> 1, strip = wave & 1;
  const int b = (net & 1) * nimg_per_set + blockIdx.x * 2 + simg;      // actor / critic read set 0, the targets set 1
  const _Float16* in = img + (size_t)b * (H * W * CIN);
  const __amdgpu_buffer_rsrc_t in_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(in) - 64, 0, H * W * CIN * 2 + 128 + 256, 0x00020000);
  const int rowbytes = W * CIN * 2;
  const int avoff = 128 + ((strip * 32 + li - P) * CIN + 8 * lj) * 2;
  const unsigned badr = (unsigned)(size_t)lds + (lj * 16 + li) * 16;
  const float bias = (float)(li + 1) * 0.01f;
  const f32x4 bias4 = {bias, bias, bias, bias};
  const float inv = 1.0f / 4096.0f;

  u32x4 av[NCH][XT];
  auto load_a = [&](int ch, int q) {
#pragma unroll
    for (int m = 0; m < XT; ++m) av[ch][m] = __builtin_amdgcn_raw_buffer_load_b128(in_rs, avoff + (m * 16 * CIN + RK * ch) * 2, q * rowbytes, 0);
  };
  f32x4 acc[KS][XT];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int m = 0; m < XT; ++m) acc[s][m] = bias4;
  float prevv[XT][2] = {{0.f, 0.f}, {0.f, 0.f}};
  int prevc[XT][2] = {{0, 0}, {0, 0}};
  const bool act = li < NO;
  const int obase = ((b * 4 + net) % (2 * nimg_per_set)) * (H / 2) * (W / 2) * NO;      // (some image's pooled plane: the probe only needs the traffic)

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) load_a(ch, 0);
  for (int q0 = 0; q0 < H + P + 3; q0 += KS) {
#pragma unroll
    for (int sq = 0; sq < KS; ++sq) {
      const int q = q0 + sq;                           // input row (rows H .. H + P - 1 load 0: the descriptor's range)
      if (q >= H + P) break;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        u32x4 a0 = av[ch][0], a1 = av[ch][1];
        if (ch == NCH - 1) {                           // the ones slots of the shipped kernel: one and-or per tile and chunk
          a0[3] = (a0[3] & 0x0000FFFFu) | 0x3C000000u; a1[3] = (a1[3] & 0x0000FFFFu) | 0x3C000000u;
        }
#pragma unroll
        for (int pc = NPC - 1; pc >= 0; --pc) {
          f16x8 bv[KS];
#pragma unroll
          for (int ky = 0; ky < KS; ++ky)
            bv[ky] = *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((uintptr_t)(badr + (unsigned)(((ky * NCH + ch) * NPC + pc) * SLAB)));
#pragma unroll
          for (int ky = 0; ky < KS; ++ky) {
            const int s = (sq + P - ky + KS) % KS;     // output row q - ky (in padded coordinates) lives in set s
            const bool first = ky == 0 && ch == 0 && pc == NPC - 1;      // the row's first contribution: C = bias, the set restarts
            acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a0), bv[ky], first ? bias4 : acc[s][0], 0, 0, 0);
            acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a1), bv[ky], first ? bias4 : acc[s][1], 0, 0, 0);
          }
        }
        load_a(ch, q + 1);
      }
      // ---- output row y = q - P (its last contribution was ky = KS - 1) is complete in set (sq + P - (KS - 1) + KS) % KS
      const int y = q - P;
      if (y >= 0) {
        const int e = (sq + P + 1) % KS;
        float xv[XT][2]; int xc[XT][2];
#pragma unroll
        for (int m = 0; m < XT; ++m)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const float z0 = acc[e][m][2 * h], z1 = acc[e][m][2 * h + 1];
            xv[m][h] = z1 > z0 ? z1 : z0; xc[m][h] = z1 > z0 ? 1 : 0;
          }
        if (y & 1) {
          const int py = y >> 1;
#pragma unroll
          for (int m = 0; m < XT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bool lower = xv[m][h] > prevv[m][h];
              const float mx = lower ? xv[m][h] : prevv[m][h];
              const int code = (lower ? 2 + xc[m][h] : prevc[m][h]) | (mx > 0.f ? 4 : 0);
              const float pv = mx > 0.f ? mx * inv : 0.f;
              const unsigned hb = __float_as_uint(pv) & 0xFFFF0000u;
              const float r1 = pv - __uint_as_float(hb);
              const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
              const unsigned lb = __float_as_uint(r1 - __uint_as_float(mb));
              const int px = strip * 16 + m * 8 + 2 * lj + h;
              const int o = obase + (py * (W / 2) + px) * NO + li;
              if (STORES == 1 && act) {
                out[o] = pv;
                planes[o] = (unsigned short)(hb >> 16);
                planes[o + 4 * 1024 * 1024] = (unsigned short)(mb >> 16);
                planes[o + 8 * 1024 * 1024] = (unsigned short)(lb >> 16);
                codes[o] = (unsigned char)code;
              }
              if (STORES == 2) {                       // lane li takes lane li + 1's value (DPP row_shr would do): even lanes store channel pairs
                const float pv1 = __shfl_down(pv, 1); const int code1 = __shfl_down(code, 1);
                const unsigned hb1 = __float_as_uint(pv1) & 0xFFFF0000u;
                const float r11 = pv1 - __uint_as_float(hb1);
                const unsigned mb1 = __float_as_uint(r11) & 0xFFFF0000u;
                const unsigned lb1 = __float_as_uint(r11 - __uint_as_float(mb1));
                if (act && !(li & 1)) {
                  *reinterpret_cast<float2*>(out + o) = make_float2(pv, pv1);
                  *reinterpret_cast<unsigned*>(planes + o) = (hb >> 16) | hb1;
                  *reinterpret_cast<unsigned*>(planes + o + 4 * 1024 * 1024) = (mb >> 16) | mb1;
                  *reinterpret_cast<unsigned*>(planes + o + 8 * 1024 * 1024) = (lb >> 16) | (lb1 & 0xFFFF0000u);
                  *reinterpret_cast<unsigned short*>(codes + o) = (unsigned short)(code | (code1 << 8));
                }
              }
              if (STORES == 0 && pv == 123.456f) out[o] = pv + (float)code + __uint_as_float(lb);
            }
        } else {
#pragma unroll
          for (int m = 0; m < XT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) { prevv[m][h] = xv[m][h]; prevc[m][h] = xc[m][h]; }
        }
      }
    }
  }
}

template <int STORES>
void run(const char* name, const _Float16* img, const _Float16* wimg, float* out, unsigned short* planes, unsigned char* codes) {
  hipFuncSetAttribute((const void*)probe<STORES>, hipFuncAttributeMaxDynamicSharedMemorySize, WLB);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<STORES>, dim3(128, 4), dim3(256), WLB, 0, img, wimg, out, planes, codes, 256);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 50;
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<STORES>, dim3(128, 4), dim3(256), WLB, 0, img, wimg, out, planes, codes, 256);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-60s %.1f us per launch (512 workgroups, 1024 image-networks; %s)\n", name, 1e3 * ms / reps, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t nimg = 512, ipx = (size_t)H * W * CIN;
  std::vector<_Float16> himg(nimg * ipx + 4096), hw(4 * (WLB / 2));
  srand(1);
  for (auto& v : himg) v = (_Float16)((rand() % 256) / 255.0f);
  for (auto& v : hw) v = (_Float16)(((rand() % 2001) - 1000) / 16.0f);
  _Float16 *img, *wimg; float* out; unsigned short* planes; unsigned char* codes;
  hipMalloc(&img, himg.size() * 2); hipMalloc(&wimg, hw.size() * 2);
  hipMalloc(&out, 32u << 20); hipMalloc(&planes, 32u << 20); hipMalloc(&codes, 8u << 20);
  hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice); hipMemcpy(wimg, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  run<0>("row-streaming implicit GEMM, no output stores", img + 2048, wimg, out, planes, codes);
  run<1>("... one store per value (b32 / 3 x b16 / b8)", img + 2048, wimg, out, planes, codes);
  run<2>("... channel pairs per lane (b64 / 3 x b32 / b16)", img + 2048, wimg, out, planes, codes);
  return 0;
}

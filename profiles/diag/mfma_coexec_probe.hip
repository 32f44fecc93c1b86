// Do the VALU instructions of one wave run under the MFMAs of ANOTHER wave of the same SIMD when both instruction streams are BLOCKY
// (a burst of MFMAs, then a burst of VALU work -- conv_fwd_k16_kernel's row: 48 MFMAs in three bursts, ~140 VALU in three blocks), or only
// when they are finely interleaved (profiles/r02_mfma_rate_probe.txt)?  gfx950, v_mfma_f32_16x16x32_f16 (16 cycles of pipe each).
//   hipcc --offload-arch=gfx950 -O3 -o cartpoleplusplus_amd/lib/mfma_coexec_probe profiles/diag/mfma_coexec_probe.hip && cartpoleplusplus_amd/lib/mfma_coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One iteration = NM MFMAs and NV VALU instructions.
// PAT 0: all MFMAs, then all VALU (blocky).  PAT 1: after every MFMA, NV / NM VALU (interleaved).  PAT 2: blocky, and the first VALU
// instructions read the accumulators the MFMAs just wrote (the epilogue's dependence).  PAT 3: blocky, s_setprio(1) around the MFMA burst.
// PAT 4: blocky, the VALU block is ONE dependent chain (each instruction reads the previous result).
template <int NM, int NV, int PAT>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int iters, int stagger) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
  unsigned h = (blockIdx.x * 977u + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (_Float16)(((int)(h >> 16) % 4096 - 2048) / 1024.0f); };
  for (int e = 0; e < 8; ++e) { a[e] = rnd(); b[e] = rnd(); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  unsigned x[8];
  for (int i = 0; i < 8; ++i) x[i] = lane * 7 + i;
  const unsigned m0 = 0xFFFF00FFu, c0 = 0x00003C00u;
  if (stagger) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (hwid & 1u) for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (PAT == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < NM; ++s) {
      acc[s % 8] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[s % 8], 0, 0, 0);
      if (PAT == 1) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV / NM; ++v) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[(s * (NV / NM) + v) % 8]) : "v"(m0), "v"(c0));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (PAT == 3) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (PAT != 1) {
      if (PAT == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(__builtin_bit_cast(unsigned, acc[i][0])), "v"(c0));
      }
#pragma unroll
      for (int v = (PAT == 2 ? 8 : 0); v < NV; ++v) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[PAT == 4 ? 0 : v % 8]) : "v"(m0), "v"(c0));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + (float)x[i];
  if (s == 123.456f) out[0] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NM, int NV, int PAT>
void run(const char* name, int waves_per_simd, bool two_wgs, int stagger = 0) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8 * 8 * 512 + 16);
  const int iters = 4000;
  const int threads = two_wgs ? 256 : 256 * waves_per_simd, grid = two_wgs ? 256 * waves_per_simd : 256;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<NM, NV, PAT>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, stagger);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> hc(grid * threads / 64);
  hipMemcpy(hc.data(), cyc, 8 * hc.size(), hipMemcpyDeviceToHost);
  double mx = 0, sum = 0; for (auto c : hc) { mx = c > mx ? (double)c : mx; sum += (double)c; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NM, NV, PAT>), dim3(grid), dim3(threads), 0, 0, out, cyc, iters, stagger);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %d wave(s)/SIMD%s%s: %7.1f cycles per iteration and wave (mean; max %7.1f); pipe needs %d, VALU issue %d; wall %.3f ms\n", name, waves_per_simd,
         two_wgs ? " (two workgroups)" : "", stagger ? " staggered" : "", sum / hc.size() / iters, mx / iters, NM * 16 * waves_per_simd, NV * 4 * waves_per_simd, ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<16, 0, 0>("16 MFMA", w, false);
    run<0, 40, 0>("40 VALU (8 independent chains)", w, false);
    run<16, 40, 0>("blocky: 16 MFMA, then 40 VALU", w, false);
    run<16, 48, 1>("interleaved: 16 x (MFMA, 3 VALU)", w, false);
    run<16, 32, 1>("interleaved: 16 x (MFMA, 2 VALU)", w, false);
    run<16, 16, 1>("interleaved: 16 x (MFMA, 1 VALU)", w, false);
    run<16, 40, 2>("blocky, the VALU block reads the accumulators first", w, false);
    run<16, 40, 3>("blocky, s_setprio(1) around the MFMA burst", w, false);
    run<16, 40, 4>("blocky, the VALU block is one dependent chain", w, false);
    run<8, 20, 0>("blocky: 8 MFMA, then 20 VALU", w, false);
    run<48, 140, 0>("blocky: 48 MFMA, then 140 VALU (a conv1 row)", w, false);
  }
  run<16, 40, 0>("blocky: 16 MFMA, then 40 VALU", 2, true);
  run<16, 40, 0>("blocky: 16 MFMA, then 40 VALU", 2, true, 4);
  run<16, 40, 0>("blocky: 16 MFMA, then 40 VALU", 2, false, 4);
  run<48, 140, 0>("blocky: 48 MFMA, then 140 VALU (a conv1 row)", 2, true);
  run<48, 140, 0>("blocky: 48 MFMA, then 140 VALU (a conv1 row)", 2, true, 12);
  run<16, 48, 1>("interleaved: 16 x (MFMA, 3 VALU)", 2, true);
  return 0;
}

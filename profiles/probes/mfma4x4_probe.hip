// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950: (1) lane <-> element mapping, (2) sustained issue rate.
// hipcc --offload-arch=gfx950 -O3 mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void map_kernel(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ void rate_kernel(float* out, int iters, long long* cycles) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int NACC>
__global__ void rate16_kernel(float* out, int iters, long long* cycles) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
  float *a, *b, *d; long long* cyc;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024 * 64); hipMalloc(&cyc, 8);
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 * (1 + l); }       // asymmetric
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  map_kernel<<<1, 64>>>(a, b, d);
  hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: D_block[i][j] = A(lane 4blk+i) * B(lane 4blk+j), lane 4blk+j holds rows i in regs
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int blk = l / 4, j = l % 4, i = r;
    const float want = ha[4 * blk + i] * hb[4 * blk + j];
    if (hd[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, hd[l * 4 + r], want); ++bad; }
  }
  printf("mapping hypothesis (lane=4*blk+j holds D_blk[i=reg][j], A row i from lane 4*blk+i): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  if (bad) { printf("lane0 regs: %g %g %g %g; lane1: %g %g %g %g; lane4: %g %g %g %g\n", hd[0], hd[1], hd[2], hd[3], hd[4], hd[5], hd[6], hd[7], hd[16], hd[17], hd[18], hd[19]); }

  float* out; hipMalloc(&out, 4 * 1024 * 1024);
  const int iters = 2000;
  long long hc;
#define RUN(K, NACC, WAVES, FLOPS, NAME) { K<NACC><<<256 * 2, 64 * WAVES>>>(out, iters, cyc); hipDeviceSynchronize(); \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); \
    K<NACC><<<256 * 2, 64 * WAVES>>>(out, iters, cyc); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); \
    double n = (double)iters * 8 * NACC; \
    printf("%s nacc=%d waves/block=%d: %.2f cycles/MFMA/wave, chip %.1f TFLOP/s\n", NAME, NACC, WAVES, hc / n, 512.0 * WAVES * n * FLOPS / (ms * 1e-3) / 1e12); }
  RUN(rate_kernel, 1, 4, 512.0, "4x4x1_16b ");
  RUN(rate_kernel, 2, 4, 512.0, "4x4x1_16b ");
  RUN(rate_kernel, 3, 4, 512.0, "4x4x1_16b ");
  RUN(rate_kernel, 6, 4, 512.0, "4x4x1_16b ");
  RUN(rate_kernel, 6, 8, 512.0, "4x4x1_16b ");
  RUN(rate16_kernel, 4, 4, 2048.0, "16x16x4   ");
  RUN(rate16_kernel, 4, 8, 2048.0, "16x16x4   ");
  return 0;
}

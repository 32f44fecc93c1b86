// Stand-alone check + timing of conv_rs16.h (row-streaming conv1 forward, weights in registers) against conv_k16.h (the (ky,o)-ring
// kernel the test suite validates) on the cfg3 launch: 4 networks x 256 images of 64x64x18 f16, random weights and whitening tables.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cartpoleplusplus_amd/csrc -o cartpoleplusplus_amd/lib/conv1_rs16_probe profiles/diag/conv1_rs16_probe.hip
// Prints: outputs that differ bit for bit (interior / border pooled columns separately: the border columns' constant enters the
// accumulator at a different place, see conv_rs16.h), the largest difference, and the launch times of both kernels.
#include "conv_rs16.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <vector>

void cpp_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  constexpr int CIN = 18, H = 64, W = 64, B = 256, NO = 10, NN = 4;
  typedef Rs16Geom<CIN> G;
  const size_t ipx = (size_t)H * W * CIN;
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  srand(7);
  std::vector<_Float16> himg(2 * B * ipx);
  for (auto& v : himg) v = (_Float16)((rand() % 256) / 255.0f);
  // a flat region (exact pooling ties) and a near-constant channel in the first images
  for (int y = 8; y < 24; ++y) for (int x = 8; x < 40; ++x) for (int c = 0; c < CIN; ++c) himg[((size_t)y * W + x) * CIN + c] = (_Float16)(0.25f);
  std::vector<float> hw(NN * 25 * CIN * NO), hb(NN * NO), hwh(NN * 2 * CIN);
  for (auto& v : hw) v = ((rand() % 20001) - 10000) * 1e-5f;
  for (auto& v : hb) v = ((rand() % 2001) - 1000) * 1e-4f;
  for (int n = 0; n < NN; ++n)
    for (int c = 0; c < CIN; ++c) {
      const float s = 3.0f + 0.1f * c + (c == 5 ? 900.f : 0.f), mu = 0.45f + 0.005f * c;
      hwh[n * 2 * CIN + c] = (c == 7 && n == 1) ? 0.f : s;
      hwh[n * 2 * CIN + CIN + c] = (c == 7 && n == 1) ? 0.f : -mu * s;
    }
  char* arena; CK(hipMalloc(&arena, himg.size() * 2 + 8192));
  _Float16* img = reinterpret_cast<_Float16*>(arena + 4096);
  CK(hipMemset(arena, 0, himg.size() * 2 + 8192));
  CK(hipMemcpy(img, himg.data(), himg.size() * 2, hipMemcpyHostToDevice));
  float *w, *bias, *wh; CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&bias, hb.size() * 4)); CK(hipMalloc(&wh, hwh.size() * 4));
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(wh, hwh.data(), hwh.size() * 4, hipMemcpyHostToDevice));
  const size_t pooled = (size_t)B * 32 * 32 * NO;
  float* out[2]; unsigned short* planes[2]; uint8_t* codes[2];
  for (int k = 0; k < 2; ++k) {
    CK(hipMalloc(&out[k], NN * pooled * 4)); CK(hipMalloc(&planes[k], NN * 3 * pooled * 2)); CK(hipMalloc(&codes[k], NN * pooled));
    CK(hipMemset(out[k], 0xEE, NN * pooled * 4)); CK(hipMemset(planes[k], 0xEE, NN * 3 * pooled * 2)); CK(hipMemset(codes[k], 0xEE, NN * pooled));
  }
  unsigned char* recs; CK(hipMalloc(&recs, NN * G::REC_BYTES));
  cpp_ctx ctx; memset(&ctx, 0, sizeof(ctx)); ctx.stream = 0; ctx.device = 0; ctx.num_cus = 256;
  auto make = [&](int k) {
    ConvArgsN b; memset(&b, 0, sizeof(b)); b.n = NN;
    for (int n = 0; n < NN; ++n) {
      ConvArgs& a = b.a[n];
      a.in = img + (size_t)(n >= 2 ? B : 0) * ipx; a.in_bstride = (long)ipx;
      a.scale = wh + n * 2 * CIN; a.shift = a.scale + CIN;
      a.w = w + n * 25 * CIN * NO; a.bias = bias + n * NO;
      a.out = n < 2 ? out[k] + n * pooled : nullptr; a.out_bstride = 32 * 32 * NO; a.out_amax = n < 2 ? codes[k] + n * pooled : nullptr;
      a.out_b16 = planes[k] + (size_t)n * 3 * pooled; a.out_b16_plane = (long)pooled;
      a.B = B; a.H = H; a.W = W; a.nout = NO; a.cin_rt = CIN;
      a.wimg = recs + n * G::REC_BYTES;
    }
    return b;
  };
  ConvArgsN ring = make(0), rs = make(1);
#ifdef RS16_TIMELINE
  unsigned long long* tl; CK(hipMalloc(&tl, 2048 * 4 * 8)); CK(hipMemset(tl, 0, 2048 * 4 * 8));
  for (int n = 0; n < NN; ++n) rs.a[n].partial = reinterpret_cast<float*>(tl);
#endif
  Conv1ImageArgsN ia; memset(&ia, 0, sizeof(ia)); ia.n = NN;
  for (int n = 0; n < NN; ++n) ia.a[n] = Conv1ImageArgs{rs.a[n].w, rs.a[n].bias, rs.a[n].scale, rs.a[n].shift, 0.f, NO, recs + n * G::REC_BYTES, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, 0.f, nullptr, nullptr};
  const int ilds = Rs16ImageLds<CIN>::BYTES;
  CK(hipFuncSetAttribute((const void*)conv1_image_kernel<CIN>, hipFuncAttributeMaxDynamicSharedMemorySize, ilds));
  auto run_ring = [&]() { if (conv_fwd_k16_launch_t<CIN, 5, 2, 2>(&ctx, ring)) exit(2); };
  auto run_img = [&]() { hipLaunchKernelGGL(conv1_image_kernel<CIN>, dim3(NN), dim3(CONV_THREADS), ilds, 0, ia); };
  auto run_rs = [&]() { hipLaunchKernelGGL(conv_fwd_rs16_kernel<CIN>, dim3(B / 2, NN), dim3(CONV_THREADS), 0, 0, rs); };
  run_ring(); run_img(); run_rs();
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  // ---- compare
  std::vector<float> o0(NN * pooled), o1(NN * pooled);
  std::vector<unsigned short> p0(NN * 3 * pooled), p1(NN * 3 * pooled);
  std::vector<uint8_t> c0(NN * pooled), c1(NN * pooled);
  CK(hipMemcpy(o0.data(), out[0], o0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), out[1], o1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(p0.data(), planes[0], p0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), planes[1], p1.size() * 2, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c0.data(), codes[0], c0.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), codes[1], c1.size(), hipMemcpyDeviceToHost));
  long nd_int = 0, nd_bor = 0, nd_code = 0, nd_plane = 0, first = -1; double maxd = 0, maxv = 0;
  for (int n = 0; n < 2; ++n)
    for (size_t i = 0; i < pooled; ++i) {
      const size_t j = n * pooled + i;
      const int px = (int)((i / NO) % 32);
      const bool border = px == 0 || px == 31;
      if (memcmp(&o0[j], &o1[j], 4)) { (border ? nd_bor : nd_int)++; if (!border && first < 0) first = (long)j; }
      maxd = fmax(maxd, fabs((double)o0[j] - (double)o1[j])); maxv = fmax(maxv, fabs((double)o0[j]));
      if (c0[j] != c1[j] && !border) nd_code++;
    }
  // planes: every network; rebuild the f32 value from the three planes
  double maxdp = 0;
  for (int n = 0; n < NN; ++n)
    for (size_t i = 0; i < pooled; ++i) {
      float v[2];
      for (int k = 0; k < 2; ++k) {
        const std::vector<unsigned short>& p = k ? p1 : p0;
        const size_t base = (size_t)n * 3 * pooled + i;
        unsigned hb = (unsigned)p[base] << 16, mb = (unsigned)p[base + pooled] << 16, lb = (unsigned)p[base + 2 * pooled] << 16;
        float fh, fm, fl; memcpy(&fh, &hb, 4); memcpy(&fm, &mb, 4); memcpy(&fl, &lb, 4);
        v[k] = (fh + fm) + fl;
      }
      const int px = (int)((i / NO) % 32);
      if (v[0] != v[1] && px != 0 && px != 31) nd_plane++;
      maxdp = fmax(maxdp, fabs((double)v[0] - (double)v[1]));
    }
  {  // where the interior differences sit
    std::vector<long> bypy(32, 0), bypx(32, 0), byo(NO, 0), byimg(8, 0);
    for (int n = 0; n < 2; ++n)
      for (size_t i = 0; i < pooled; ++i) {
        const size_t j = n * pooled + i;
        const int o = (int)(i % NO), px = (int)((i / NO) % 32), py = (int)((i / (NO * 32)) % 32), im = (int)(i / (NO * 1024));
        if (px == 0 || px == 31) continue;
        if (memcmp(&o0[j], &o1[j], 4)) { bypy[py]++; bypx[px]++; byo[o]++; byimg[im & 7]++; }
      }
    printf("  by pooled row:"); for (int k = 0; k < 32; ++k) printf(" %ld", bypy[k]); printf("\n");
    printf("  by pooled col:"); for (int k = 0; k < 32; ++k) printf(" %ld", bypx[k]); printf("\n");
    printf("  by filter:"); for (int k = 0; k < NO; ++k) printf(" %ld", byo[k]); printf("\n");
    printf("  by image & 7:"); for (int k = 0; k < 8; ++k) printf(" %ld", byimg[k]); printf("\n");
  }
  printf("f32 pool1 of 2 networks: %ld interior / %ld border values differ bit for bit (of %zu), max |diff| %.3e at max |value| %.3f; codes differ (interior) %ld\n",
         nd_int, nd_bor, 2 * pooled, maxd, maxv, nd_code);
  printf("bf16 planes of 4 networks: %ld interior values differ, max |diff| %.3e\n", nd_plane, maxdp);
  if (first >= 0) printf("  first interior difference at %ld: ring %.9g rs16 %.9g\n", first, o0[first], o1[first]);
#ifdef RS16_TIMELINE
  {
    run_rs(); run_rs(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2048 * 4); CK(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0; for (int i = 0; i < 2048; ++i) { if (h[4 * i] < t0) t0 = h[4 * i]; if (h[4 * i + 2] > t1) t1 = h[4 * i + 2]; }
    printf("timeline: first wave start .. last wave end = %.2f us\n", (t1 - t0) / 100.0);
    // histogram of start times and durations (us)
    int hs[12] = {0}, hd[12] = {0}; double sum_d = 0, sum_setup = 0;
    for (int i = 0; i < 2048; ++i) { const double st = (h[4 * i] - t0) / 100.0, du = (h[4 * i + 2] - h[4 * i]) / 100.0; hs[(int)(st / 10) > 11 ? 11 : (int)(st / 10)]++; hd[(int)(du / 10) > 11 ? 11 : (int)(du / 10)]++; sum_d += du; sum_setup += (h[4 * i + 1] - h[4 * i]) / 100.0; }
    printf("  start time histogram (10 us bins):"); for (int k = 0; k < 12; ++k) printf(" %d", hs[k]); printf("\n  duration histogram (10 us bins):"); for (int k = 0; k < 12; ++k) printf(" %d", hd[k]);
    printf("\n  mean wave duration %.2f us, mean setup (to the row loop) %.2f us\n", sum_d / 2048, sum_setup / 2048);
    // waves per SIMD at mid time: count waves by (cu, simd) from HW_ID
    for (int i = 0; i < 8; ++i) printf("  wave %d: start %.2f loop %.2f end %.2f hwid %llx\n", i * 251, (h[4 * i * 251] - t0) / 100.0, (h[4 * i * 251 + 1] - t0) / 100.0, (h[4 * i * 251 + 2] - t0) / 100.0, h[4 * i * 251 + 3]);
  }
#endif
  // ---- time
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto&& f) {
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.2f us per launch\n", name, 1e3 * ms / reps);
  };
  time("conv_fwd_k16_kernel<18,5,2,2> (ring)", run_ring);
  time("conv1_image_kernel<18> (4 networks)", run_img);
  time("conv_fwd_rs16_kernel<18>", run_rs);
  time("image + rs16", [&]() { run_img(); run_rs(); });
  return 0;
}

// gfx950: the same question as store_hazard_probe.hip for LDS stores -- does a VALU write of a ds_write's DATA registers right behind it
// change what lands in LDS?  (LLVM has no such hazard for this target; the row-streaming kernels issue ds_write_b128 / b64 by the thousand.)
// build on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/p profiles/diag/lds_store_hazard_probe.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define POISON 0xDEADBEEFu

#define LDS_KERNEL(NAME, OP, REGS, NOPS)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters) {                                                     \
    __shared__ __attribute__((aligned(16))) unsigned lds[256 * 4];                                                            \
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;                                                                      \
    const unsigned la = (unsigned)(unsigned long long)(lds + threadIdx.x * 4);                                                \
    for (int it = 0; it < iters; ++it) {                                                                                       \
      const unsigned val = gid * 977u + (unsigned)it;                                                                          \
      unsigned r0, r1, r2, r3;                                                                                                 \
      asm volatile("v_mov_b32 v100, %5\n v_add_u32 v101, 1, %5\n v_add_u32 v102, 2, %5\n v_add_u32 v103, 3, %5\n s_nop 4\n"    \
                   OP " %4, " REGS "\n" NOPS "v_mov_b32 v101, %6\n"                                                            \
                   "s_waitcnt lgkmcnt(0)\n ds_read_b128 v[104:107], %4\n s_waitcnt lgkmcnt(0)\n"                               \
                   "v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n v_mov_b32 %2, v106\n v_mov_b32 %3, v107\n"                       \
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(la), "v"(val), "v"(POISON)                                   \
                   : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "memory");                               \
      unsigned* o = out + ((size_t)gid * iters + it) * 4;                                                                      \
      o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;                                                                              \
    }                                                                                                                          \
  }
LDS_KERNEL(l_128_0, "ds_write_b128", "v[100:103]", "")
LDS_KERNEL(l_128_1, "ds_write_b128", "v[100:103]", "s_nop 0\n")
LDS_KERNEL(l_64_0, "ds_write_b64", "v[100:101]", "")
LDS_KERNEL(l_64_1, "ds_write_b64", "v[100:101]", "s_nop 0\n")

typedef void (*kern_t)(unsigned*, int);
int main() {
  const int blocks = 4096, iters = 16;
  const size_t words = (size_t)blocks * 256 * iters * 4;
  unsigned* d; hipMalloc(&d, words * 4);
  std::vector<unsigned> h(words);
  struct { const char* name; kern_t k; int width; } cases[] = {{"ds_write_b128 +0", l_128_0, 4}, {"ds_write_b128 +1", l_128_1, 4}, {"ds_write_b64  +0", l_64_0, 2}, {"ds_write_b64  +1", l_64_1, 2}};
  for (auto& c : cases) {
    long poisoned = 0, wrong = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0, words * 4);
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, iters);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
      for (size_t g = 0; g < (size_t)blocks * 256; ++g)
        for (int it = 0; it < iters; ++it) {
          const size_t base = (g * iters + it) * 4;
          const unsigned val = (unsigned)g * 977u + (unsigned)it;
          for (int e = 0; e < c.width; ++e) {
            if (h[base + e] == POISON) ++poisoned; else if (h[base + e] != val + (unsigned)e) ++wrong;
          }
        }
    }
    printf("LDSHAZARD %-18s poisoned %ld  otherwise wrong %ld  of %ld words\n", c.name, poisoned, wrong, (long)blocks * 256 * iters * c.width * 3);
  }
  hipFree(d);
  return 0;
}

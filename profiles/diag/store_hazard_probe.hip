// gfx950: does a VALU write of a buffer store's DATA registers right behind the store change what is stored?
// One kernel per (store width, soffset kind, wait states K between the store and the overwrite): every lane stores known values to its own
// addresses, overwrites one data register with a poison K wait states later, and the host counts poisoned words in memory.
// build: hipcc --offload-arch=gfx950 -O2 -o store_hazard_probe profiles/diag/store_hazard_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
#define POISON 0xDEADBEEFu

#define NOP_0
#define NOP_1 "s_nop 0\n"
#define NOP_2 "s_nop 1\n"
#define NOP_3 "s_nop 2\n"
#define NOP_4 "s_nop 3\n"

// WIDTH: 2 / 3 / 4 dwords; SOFF: "%3" (an SGPR) or "0"; the data registers are v[100:103], the poisoned one is v101
#define PROBE_KERNEL(NAME, STOREOP, REGS, SOFF, NOPS)                                                                        \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, int stride_bytes) {                                  \
    const i32x4 rsrc = {(int)(unsigned)(unsigned long long)out, (int)(((unsigned long long)out >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000}; \
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;                                                                      \
    for (int it = 0; it < iters; ++it) {                                                                                       \
      const int voff = (int)((gid * (unsigned)iters + (unsigned)it) * 16u);                                                    \
      const unsigned val = gid * 977u + (unsigned)it;                                                                          \
      const int soff = __builtin_amdgcn_readfirstlane(stride_bytes);                                                           \
      asm volatile("v_mov_b32 v100, %1\n v_add_u32 v101, 1, %1\n v_add_u32 v102, 2, %1\n v_add_u32 v103, 3, %1\n s_nop 4\n"    \
                   STOREOP " " REGS ", %0, %2, " SOFF " offen\n" NOPS                                                           \
                   "v_mov_b32 v101, %4\n s_nop 4\n"                                                                            \
                   :: "v"(voff), "v"(val), "s"(rsrc), "s"(soff), "v"(POISON) : "v100", "v101", "v102", "v103", "memory");     \
    }                                                                                                                          \
  }

#define FAMILY(W, OP, REGS)                                                       \
  PROBE_KERNEL(k_##W##_s_0, OP, REGS, "%3", NOP_0) PROBE_KERNEL(k_##W##_s_1, OP, REGS, "%3", NOP_1)   \
  PROBE_KERNEL(k_##W##_s_2, OP, REGS, "%3", NOP_2) PROBE_KERNEL(k_##W##_s_3, OP, REGS, "%3", NOP_3)   \
  PROBE_KERNEL(k_##W##_s_4, OP, REGS, "%3", NOP_4)                                                       \
  PROBE_KERNEL(k_##W##_i_0, OP, REGS, "0", NOP_0) PROBE_KERNEL(k_##W##_i_1, OP, REGS, "0", NOP_1) PROBE_KERNEL(k_##W##_i_2, OP, REGS, "0", NOP_2)

FAMILY(2, "buffer_store_dwordx2", "v[100:101]")
FAMILY(3, "buffer_store_dwordx3", "v[100:102]")
FAMILY(4, "buffer_store_dwordx4", "v[100:103]")

typedef void (*kern_t)(unsigned*, int, int);
struct Case { const char* name; kern_t k; int width; };

int main() {
  const int blocks = 4096, iters = 32;
  const size_t words = (size_t)blocks * 256 * iters * 4;
  unsigned* d; hipMalloc(&d, words * 4);
  std::vector<unsigned> h(words);
  Case cases[] = {
#define C(W) {"x" #W " sgpr-soffset +0", k_##W##_s_0, W}, {"x" #W " sgpr-soffset +1", k_##W##_s_1, W}, {"x" #W " sgpr-soffset +2", k_##W##_s_2, W}, \
             {"x" #W " sgpr-soffset +3", k_##W##_s_3, W}, {"x" #W " sgpr-soffset +4", k_##W##_s_4, W}, \
             {"x" #W " imm-soffset  +0", k_##W##_i_0, W}, {"x" #W " imm-soffset  +1", k_##W##_i_1, W}, {"x" #W " imm-soffset  +2", k_##W##_i_2, W},
    C(2) C(3) C(4)
  };
  for (const Case& c : cases) {
    long poisoned = 0, wrong = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0, words * 4);
      hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, iters, 0);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, words * 4, hipMemcpyDeviceToHost);
      for (size_t g = 0; g < (size_t)blocks * 256; ++g)
        for (int it = 0; it < iters; ++it) {
          const size_t base = (g * iters + it) * 4;
          const unsigned val = (unsigned)g * 977u + (unsigned)it;
          for (int e = 0; e < c.width; ++e) {
            if (h[base + e] == POISON) ++poisoned;
            else if (h[base + e] != val + (unsigned)e) ++wrong;
          }
        }
    }
    printf("HAZARD %-24s poisoned %ld  otherwise wrong %ld  of %ld stored words\n", c.name, poisoned, wrong, (long)blocks * 256 * iters * c.width * 3);
  }
  hipFree(d);
  return 0;
}

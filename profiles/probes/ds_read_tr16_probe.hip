#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(short* out, const short* in) {
  extern __shared__ __attribute__((aligned(16))) short lds[];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // every lane supplies the address of 4 contiguous halves: here a [64 lanes][4] row-major image
  __attribute__((address_space(3))) v4s* p = (__attribute__((address_space(3))) v4s*)(uintptr_t)(uint32_t)(uintptr_t)(lds + lane * 4);
  v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  std::vector<short> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short *din, *dout; hipMalloc(&din, 8192); hipMalloc(&dout, 512);
  hipMemcpy(din, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, dout, din);
  std::vector<short> o(256); hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf("\n"); }
  return 0;
}

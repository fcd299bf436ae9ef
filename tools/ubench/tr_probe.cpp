#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int lane = threadIdx.x;
  // mode 0: lane address = lane * 8 bytes (4 elements each)
  // mode 1: lane address: 16-lane group g reads [4 rows][16 cols] block: row pitch 64 elements; lane a=4j+jl -> row j, col 4 jl
  uint32_t addr;
  if (mode == 0) addr = lane * 8;
  else { int a = lane & 15, g = lane >> 4; addr = (g * 4 * 64 + (a >> 2) * 64 + (a & 3) * 4) * 2; }
  addr += (uint32_t)(uintptr_t)lds;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(uintptr_t)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, 1, 64, 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}

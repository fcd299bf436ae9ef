// What does the "lane = row" access pattern of the FeatureEnhancer chains (csrc/fe_chain.hip fc_load_row / fc_store_row:
// every lane moves 16-byte pieces of ITS OWN 512-byte row, a wave instruction touches 32 rows x 32 bytes) cost against a
// fully coalesced copy of the same bytes?  Pure copies, same grid shape as the chains (256 blocks x 8 waves, 32-row tiles).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/rowpat_ubench.cpp -o build/rowpat_ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// NIN input matrices read, NOUT written (each [rows][128] fp32), lane = row
template <int NIN, int NOUT>
__global__ __launch_bounds__(512, 1) void rowlane_kernel(const float* __restrict__ in, float* __restrict__ out, int ntiles, long mat) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    float4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < NIN; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(in + m * mat + row * 128 + 8 * i + 4 * lh);
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
      }
#pragma unroll
    for (int m = 0; m < NOUT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(out + m * mat + row * 128 + 8 * i + 4 * lh) = acc[i];
  }
}
// the same bytes, every wave instruction = 1 KB contiguous (2 rows)
template <int NIN, int NOUT>
__global__ __launch_bounds__(512, 1) void coalesced_kernel(const float* __restrict__ in, float* __restrict__ out, int ntiles, long mat) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t base = (size_t)t * 32 * 128;
    float4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < NIN; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(in + m * mat + base + (size_t)i * 256 + lane * 4);
        acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
      }
#pragma unroll
    for (int m = 0; m < NOUT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(out + m * mat + base + (size_t)i * 256 + lane * 4) = acc[i];
  }
}
template <class F> static float timeit(F fn) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) fn();
  std::vector<float> ts;
  for (int i = 0; i < 10; ++i) { CK(hipEventRecord(a, 0)); for (int r = 0; r < 5; ++r) fn(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 200.f); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
int main() {
  const long rows = 131072, mat = rows * 128;
  float *in, *out; CK(hipMalloc(&in, 3 * mat * 4)); CK(hipMalloc(&out, 3 * mat * 4)); CK(hipMemset(in, 0, 3 * mat * 4));
  const int nt = rows / 32;
#define RUN(K, NI, NO, NB) { float t = timeit([&] { hipLaunchKernelGGL((K<NI, NO>), dim3(NB), 512, 0, 0, in, out, nt, mat); }); \
    printf("%-18s in %d out %d blocks %4d: %7.1f us  %.2f TB/s\n", #K, NI, NO, NB, t, (NI + NO) * (double)mat * 4 / t * 1e-6); }
  RUN(rowlane_kernel, 2, 2, 256) RUN(coalesced_kernel, 2, 2, 256)
  RUN(rowlane_kernel, 2, 2, 512) RUN(coalesced_kernel, 2, 2, 512)
  RUN(rowlane_kernel, 3, 2, 256) RUN(coalesced_kernel, 3, 2, 256)
  RUN(rowlane_kernel, 1, 1, 256) RUN(coalesced_kernel, 1, 1, 256)
  RUN(rowlane_kernel, 2, 0, 256) RUN(coalesced_kernel, 2, 0, 256)
  RUN(rowlane_kernel, 0, 2, 256) RUN(coalesced_kernel, 0, 2, 256)
  RUN(rowlane_kernel, 2, 2, 1024) RUN(coalesced_kernel, 2, 2, 1024)
  return 0;
}

// Standalone micro-benchmark + correctness check of the linear weight-gradient kernels.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-pass-failed tools/ubench/lwgrad_ubench.cpp -o build/lwgrad_ubench
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../fudanocr_amd/csrc/linear_wgrad.hip"
#include "../../fudanocr_amd/csrc/conv_bx3.hip"

extern "C" void focr_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
extern "C" int focr_get_precision(void) { return 1; }
extern "C" int focr_get_tuning(int) { return 1; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, uint32_t seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = hash32((uint32_t)i * 2654435761u + seed); p[i] = scale * ((h >> 8) * (1.f / 8388608.f) - 1.f); }
}
static float* dalloc(long n, uint32_t seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), 256, 0, 0, p, n, seed, scale);
  return p;
}
// one block per (co, k): double accumulation over the rows
__global__ void ref_kernel(const float* X, const float* dY, double* dW, double* db, long M, int K, int Cout, int ldx, int ldd) {
  __shared__ double red[256];
  const int co = blockIdx.y, k = blockIdx.x;
  double s = 0.0, sb = 0.0;
  for (long p = threadIdx.x; p < M; p += 256) { double d = dY[p * ldd + co]; s += d * X[p * ldx + k]; sb += d; }
  red[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) dW[(size_t)co * K + k] = red[0];
  __syncthreads();
  if (k == 0) { red[threadIdx.x] = sb; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) db[co] = red[0]; }
}
template <class F> static float timeit(F fn, int iters = 12) {
  const int REP = 10;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) fn();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a, 0)); for (int r = 0; r < REP; ++r) fn(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f / REP); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
struct Shape { const char* name; long M; int K, Cout, ldx, ldd; };

int main() {
  const Shape shapes[] = {{"proj 128->128", 131072, 128, 128, 128, 128}, {"qkv 128->384", 131072, 128, 384, 128, 384},
                          {"b64 128->128", 65536, 128, 128, 128, 128},   {"small rows", 4096, 128, 128, 128, 128},
                          {"ragged rows", 131072 - 48, 128, 128, 128, 128}, {"strided", 32768, 128, 128, 256, 384},
                          {"out 128->64", 131072, 128, 64, 128, 64},     {"out 128->64 b64", 65536, 128, 64, 128, 64},
                          {"out 128->64 ragged", 131072 - 80, 128, 64, 128, 64}, {"out 128->64 small", 2048, 128, 64, 128, 64}};
  for (const Shape& s : shapes) {
    float* X = dalloc(s.M * s.ldx, 11, 1.f);
    float* dY = dalloc(s.M * s.ldd, 23, 0.01f);
    const long nw = (long)s.Cout * s.K;
    double *rw, *rb; CK(hipMalloc(&rw, nw * 8)); CK(hipMalloc(&rb, s.Cout * 8));
    hipLaunchKernelGGL(ref_kernel, dim3(s.K, s.Cout), 256, 0, 0, X, dY, rw, rb, s.M, s.K, s.Cout, s.ldx, s.ldd);
    std::vector<double> hrw(nw), hrb(s.Cout);
    CK(hipMemcpy(hrw.data(), rw, nw * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hrb.data(), rb, s.Cout * 8, hipMemcpyDeviceToHost));
    double mw = 0, mb = 0; for (double v : hrw) mw = std::max(mw, fabs(v)); for (double v : hrb) mb = std::max(mb, fabs(v));
    float *dw, *db; CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&db, s.Cout * 4));
    auto check = [&](const char* tag) {
      std::vector<float> hw(nw), hb(s.Cout);
      CK(hipMemcpy(hw.data(), dw, nw * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), db, s.Cout * 4, hipMemcpyDeviceToHost));
      double ew = 0, eb = 0;
      for (long i = 0; i < nw; ++i) { double d = fabs(hw[i] - hrw[i]); if (!(d <= ew)) ew = d; }
      for (int i = 0; i < s.Cout; ++i) { double d = fabs(hb[i] - hrb[i]); if (!(d <= eb)) eb = d; }
      printf("  %-8s rel err dW %.2e  db %.2e\n", tag, ew / mw, eb / mb);
    };
    printf("%s: M=%ld K=%d Cout=%d ldx=%d ldd=%d  (%.1f MB)\n", s.name, s.M, s.K, s.Cout, s.ldx, s.ldd, s.M * (s.K + s.Cout) * 4 / 1e6);
    // new kernel
    if (focr_linear_wgrad_eligible(s.M, s.K, s.Cout, s.ldx, s.ldd)) {
      long nws = focr_linear_wgrad_ws_floats(s.M, s.K, s.Cout);
      float* ws; CK(hipMalloc(&ws, nws * 4));
      CK(hipMemset(dw, 0xff, nw * 4)); CK(hipMemset(db, 0xff, s.Cout * 4));
      focr_linear_wgrad(X, dY, dw, db, ws, nws, s.M, s.K, s.Cout, s.ldx, s.ldd, 0, 0);
      CK(hipDeviceSynchronize());
      check("stream");
      // accumulate semantics: a second call with accumulate=1 doubles the result
      float t = timeit([&] { focr_linear_wgrad(X, dY, dw, db, ws, nws, s.M, s.K, s.Cout, s.ldx, s.ldd, 0, 0); });
      printf("  stream   %.1f us  (%.0f GB/s)\n", t, s.M * (s.K + s.Cout) * 4 / t / 1e3);
      CK(hipFree(ws));
    } else printf("  stream   not eligible\n");
    if (s.ldx == s.K) {
      long nws = focr_conv_wgrad_bx3_ws_floats((int)s.M, s.K, s.Cout, s.K);
      float* ws = nullptr; if (nws) CK(hipMalloc(&ws, nws * 4));
      CK(hipMemset(dw, 0, nw * 4)); CK(hipMemset(db, 0, s.Cout * 4));
      focr_conv_wgrad_bx3(X, dY, dw, db, ws, nws, 1, 1, (int)s.M, s.K, 1, (int)s.M, s.Cout, 1, 1, 0, 0, (int)s.M, s.ldd, s.ldx, 1, (int)s.M, 0);
      CK(hipDeviceSynchronize());
      check("old");
      float t = timeit([&] { focr_conv_wgrad_bx3(X, dY, dw, db, ws, nws, 1, 1, (int)s.M, s.K, 1, (int)s.M, s.Cout, 1, 1, 0, 0, (int)s.M, s.ldd, s.ldx, 1, (int)s.M, 0); });
      printf("  old      %.1f us\n", t);
      if (ws) CK(hipFree(ws));
    }
    CK(hipFree(X)); CK(hipFree(dY)); CK(hipFree(rw)); CK(hipFree(rb)); CK(hipFree(dw)); CK(hipFree(db));
  }
  return 0;
}

// Standalone micro-benchmark of the attention kernels (no torch): variant A/B in one process, outputs compared.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench/attn_ubench.cpp -o build/attn_ubench
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>
#define FOCR_UBENCH_BWD1W 1
#define FOCR_UBENCH_PLANES 1
#include "../../fudanocr_amd/csrc/attention.hip"
#include "../../fudanocr_amd/csrc/attention_bx3.hip"

extern "C" void focr_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
static int g_prec = 2;
extern "C" int focr_get_precision(void) { return g_prec; }
static int g_tune[FOCR_TUNING_COUNT] = {1, 1, 1, 1, 0};
extern "C" int focr_get_tuning(int key) { return g_tune[key]; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void fill_kernel(float* p, long n, uint32_t seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = hash32((uint32_t)i * 2654435761u + seed); p[i] = scale * ((h >> 8) * (1.f / 8388608.f) - 1.f); }
}
// packed [rows][q | k | v]: q := |q|, k := -|k|
__global__ void sign_kernel(float* p, long rows, int ld, int d) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const int c = (int)(i % ld);
  if (c < d) p[i] = fabsf(p[i]);
  else if (c < 2 * d) p[i] = -fabsf(p[i]);
}
static float* dalloc(long n, uint32_t seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  if (seed) hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), 256, 0, 0, p, n, seed, scale);
  return p;
}
template <class F> static float timeit(F fn, int iters = 10) {
  const int REP = 5;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) fn();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a, 0)); for (int r = 0; r < REP; ++r) fn(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f / REP); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
static double maxdiff(const float* da, const float* db, long n, double* mx) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost));
  double m = 0, r = 0;
  for (long i = 0; i < n; ++i) { double d = fabs((double)a[i] - b[i]); if (!(d <= m)) m = d; r = std::max(r, (double)fabsf(a[i])); }
  *mx = r;
  return m;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, H = 4, N = 1024, D = 128;
  const int V0 = argc > 2 ? atoi(argv[2]) : 1, V1 = argc > 3 ? atoi(argv[3]) : 2;      // forward variants compared
  const long n = (long)B * N * D;
  float *q = dalloc(n, 1, 2.f), *k = dalloc(n, 2, 2.f), *v = dalloc(n, 3, 1.f), *dO = dalloc(n, 4, 1.f);
  float *o0 = dalloc(n, 0, 0), *o1 = dalloc(n, 0, 0), *lse0 = dalloc((long)B * H * N, 0, 0), *lse1 = dalloc((long)B * H * N, 0, 0);
  float *dq0 = dalloc(n, 0, 0), *dk0 = dalloc(n, 0, 0), *dv0 = dalloc(n, 0, 0), *dq1 = dalloc(n, 0, 0), *dk1 = dalloc(n, 0, 0), *dv1 = dalloc(n, 0, 0);
  float* work = dalloc((long)B * H * N, 0, 0);
  uint32_t* mask; CK(hipMalloc(&mask, (size_t)B * H * N * (N / 32) * 4));
  const float scale = 1.f / sqrtf(32.f);
  const double fl = 4.0 * B * H * (double)N * N * 32;
  const bool only_b1 = argc > 4 && !strcmp(argv[4], "b1");      // single-pass backward section only
  if (getenv("FOCR_UB_MASKV")) g_tune[FOCR_TUNE_ATTN_FWD_MASK] = atoi(getenv("FOCR_UB_MASKV"));
  if (argc > 4 && !strcmp(argv[4], "fm")) {
    // round 5: keep-word schedule of the 256-query forward (tuning key 4): kernel alone on pre-drawn bits, packed
    // 1536-byte-pitch operands as in the step, interleaved A B A B A B, outputs must be bit-identical; p = 0 for reference
    float* qkv = dalloc(3 * n, 7, 1.5f);
    const int ldp = 3 * D;
    CK(focr_attention_dropout_mask(mask, B, H, N, 0.1f, 1234, 0) ? hipErrorUnknown : hipSuccess);
    g_tune[FOCR_TUNE_ATTN_FWD_VARIANT] = 1;
    float tv[3] = {1e9f, 1e9f, 1e9f}, tp0[3] = {1e9f, 1e9f, 1e9f};
    float *o2 = dalloc(n, 0, 0), *lse2 = dalloc((long)B * H * N, 0, 0);
    for (int rep = 0; rep < 3; ++rep) {
      for (int mv = 0; mv < 3; ++mv) {
        g_tune[FOCR_TUNE_ATTN_FWD_MASK] = mv;
        float* o = mv == 0 ? o0 : mv == 1 ? o1 : o2; float* ls = mv == 0 ? lse0 : mv == 1 ? lse1 : lse2;
        int rc = focr_attention_fwd_premasked(qkv, qkv + D, qkv + 2 * D, o, ls, mask, B, H, N, ldp, D, scale, 0.1f, 0);
        if (rc) { printf("premasked forward failed %d\n", rc); return 1; }
        tv[mv] = std::min(tv[mv], timeit([&]() { focr_attention_fwd_premasked(qkv, qkv + D, qkv + 2 * D, o, ls, mask, B, H, N, ldp, D, scale, 0.1f, 0); }, 8));
        tp0[mv] = std::min(tp0[mv], timeit([&]() { focr_attention_fwd(qkv, qkv + D, qkv + 2 * D, dq0, work, mask, B, H, N, ldp, D, scale, 0.f, 1234, 0); }, 8));
      }
    }
    CK(hipDeviceSynchronize());
    double mx, e = maxdiff(o0, o1, n, &mx), mx2, e2 = maxdiff(lse0, lse1, (long)B * H * N, &mx2);
    double e3 = maxdiff(o0, o2, n, &mx), e4 = maxdiff(lse0, lse2, (long)B * H * N, &mx2);
    printf("fwd keep-word schedule (packed operands, premasked, p = 0.1): MV0 %7.1f us  MV1 %7.1f us  MV2 %7.1f us | p = 0: %7.1f / %7.1f / %7.1f us\n"
           "   MV1 vs MV0: max|dO| %.2e of %.2e, max|dLSE| %.2e;  MV2 vs MV0: max|dO| %.2e, max|dLSE| %.2e of %.2e\n",
           tv[0], tv[1], tv[2], tp0[0], tp0[1], tp0[2], e, mx, e2, e3, e4, mx2);
    // robustness of the relative-max form (MV2): (a) large scores of both signs, (b) EVERY score of every row far below
    // zero (q >= 0, k <= 0): the first key group must move the reference down to the true maximum
    for (int cas = 0; cas < 2; ++cas) {
      float* qs = dalloc(3 * n, 99, 12.f);
      if (cas == 1) hipLaunchKernelGGL(sign_kernel, dim3((unsigned)((B * (long)N * 3 * D + 255) / 256)), 256, 0, 0, qs, (long)B * N, 3 * D, D);
      for (int mv : {0, 2}) {
        g_tune[FOCR_TUNE_ATTN_FWD_MASK] = mv;
        focr_attention_fwd_premasked(qs, qs + D, qs + 2 * D, mv ? o2 : o0, mv ? lse2 : lse0, mask, B, H, N, ldp, D, scale, 0.1f, 0);
      }
      CK(hipDeviceSynchronize());
      double a_ = maxdiff(o0, o2, n, &mx), b_ = maxdiff(lse0, lse2, (long)B * H * N, &mx2);
      printf("   %s: MV2 vs MV0 max|dO| %.2e of %.2e, max|dLSE| %.2e of %.2e\n",
             cas ? "all scores negative (q >= 0, k <= 0, |.| <= 12)" : "large scores (|q|, |k| <= 12)", a_, mx, b_, mx2);
      CK(hipFree(qs));
    }
    return 0;
  }
  for (float p : {0.1f, 0.0f}) {
    float t0 = 1e9f, t1 = 1e9f;
    if (only_b1) {
      g_tune[FOCR_TUNE_ATTN_FWD_VARIANT] = 1;
      focr_attention_fwd(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, p, 1234, 0);
      g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 1;
      focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq1, dk1, dv1, work, B, H, N, D, D, scale, p, 0);
    }
    if (!only_b1) {
    for (int rep = 0; rep < 3; ++rep) {                 // A B A B A B: best of three each (box clocks drift)
      g_tune[FOCR_TUNE_ATTN_FWD_VARIANT] = V0;
      focr_attention_fwd(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, p, 1234, 0);
      t0 = std::min(t0, timeit([&]() { focr_attention_fwd(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, p, 1234, 0); }, 6));
      g_tune[FOCR_TUNE_ATTN_FWD_VARIANT] = V1;
      focr_attention_fwd(q, k, v, o1, lse1, mask, B, H, N, D, D, scale, p, 1234, 0);
      t1 = std::min(t1, timeit([&]() { focr_attention_fwd(q, k, v, o1, lse1, mask, B, H, N, D, D, scale, p, 1234, 0); }, 6));
    }
    float tm = p > 0 ? timeit([&]() { hipLaunchKernelGGL(attn_mask_kernel, dim3(262144), 256, 0, 0, mask, (long)B * H * N * (N / 32), p, (uint64_t)1234); }) : 0.f;
    CK(hipDeviceSynchronize());
    double mx, e = maxdiff(o0, o1, n, &mx), mx2, e2 = maxdiff(lse0, lse1, (long)B * H * N, &mx2);
    printf("fwd p=%.1f: variantA %7.1f us (%5.0f TF)  variant1 %7.1f us (%5.0f TF)  [mask kernel alone %6.1f us]  max|dO| %.2e of %.2e, max|dLSE| %.2e\n",
           p, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, tm, e, mx, e2);
    // backward (variants of the backward are compared the same way once they exist)
    g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 0;
    focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq0, dk0, dv0, work, B, H, N, D, D, scale, p, 0);
    float b0 = timeit([&]() { focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq0, dk0, dv0, work, B, H, N, D, D, scale, p, 0); });
    g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 1;
    focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq1, dk1, dv1, work, B, H, N, D, D, scale, p, 0);
    float b1 = timeit([&]() { focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq1, dk1, dv1, work, B, H, N, D, D, scale, p, 0); });
    CK(hipDeviceSynchronize());
    double m1, m2, m3;
    double eq = maxdiff(dq0, dq1, n, &m1), ek = maxdiff(dk0, dk1, n, &m2), ev = maxdiff(dv0, dv1, n, &m3);
    printf("bwd p=%.1f: variant0 %7.1f us  variant1 %7.1f us  max diff dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e\n", p, b0, b1, eq, m1, ek, m2, ev, m3);
    fflush(stdout);
    }
    {   // single-pass backward (variant 2) against the two-pass result of variant 1 (dq0 / dk0 / dv0 hold variant 0, dq1.. variant 1)
      float *dq2 = dalloc(n, 0, 0), *dk2 = dalloc(n, 0, 0), *dv2 = dalloc(n, 0, 0);
      CK(hipMemset(dq2, 0xff, n * 4)); CK(hipMemset(dk2, 0xff, n * 4)); CK(hipMemset(dv2, 0xff, n * 4));
      g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 2;
      int rc2 = focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0);
      CK(hipDeviceSynchronize());
      if (rc2) { printf("single-pass bwd failed %d\n", rc2); return 1; }
      float t2 = 1e9f, t1b = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 1;
        t1b = std::min(t1b, timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq1, dk1, dv1, work, B, H, N, D, D, scale, p, 0); }, 6));
        g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 2;
        t2 = std::min(t2, timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0); }, 6));
      }
      CK(hipDeviceSynchronize());
      double n1, n2, n3;
      double fq = maxdiff(dq1, dq2, n, &n1), fk = maxdiff(dk1, dk2, n, &n2), fv = maxdiff(dv1, dv2, n, &n3);
      printf("bwd1 p=%.1f: two-pass (no prep) %7.1f us  single-pass %7.1f us  max diff dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e\n", p, t1b, t2, fq, n1, fk, n2, fv, n3);
      fflush(stdout);
      // one wave per SIMD variant (tuning value 3) against the same two-pass result
      CK(hipMemset(dq2, 0xff, n * 4)); CK(hipMemset(dk2, 0xff, n * 4)); CK(hipMemset(dv2, 0xff, n * 4));
      g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 3;
      focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0);
      CK(hipDeviceSynchronize());
      float t3 = 1e9f;
      for (int rep = 0; rep < 3; ++rep)
        t3 = std::min(t3, timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0); }, 6));
      CK(hipDeviceSynchronize());
      fq = maxdiff(dq1, dq2, n, &n1); fk = maxdiff(dk1, dk2, n, &n2); fv = maxdiff(dv1, dv2, n, &n3);
      printf("bwd1w p=%.1f: one wave per SIMD %7.1f us  max diff dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e\n", p, t3, fq, n1, fk, n2, fv, n3);
      fflush(stdout);
      // precision mode 3: dP = dO V^T as a single bf16 product (template flag DP1), against the same two-pass result
      CK(hipMemset(dq2, 0xff, n * 4)); CK(hipMemset(dk2, 0xff, n * 4)); CK(hipMemset(dv2, 0xff, n * 4));
      g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 2;
      g_prec = 3;
      focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0);
      CK(hipDeviceSynchronize());
      float t4 = 1e9f, t2b = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        g_prec = 2;
        t2b = std::min(t2b, timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq0, dk0, dv0, work, B, H, N, D, D, scale, p, 0); }, 6));
        g_prec = 3;
        t4 = std::min(t4, timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq2, dk2, dv2, work, B, H, N, D, D, scale, p, 0); }, 6));
      }
      g_prec = 2;
      CK(hipDeviceSynchronize());
      fq = maxdiff(dq1, dq2, n, &n1); fk = maxdiff(dk1, dk2, n, &n2); fv = maxdiff(dv1, dv2, n, &n3);
      printf("bwd1 DP1 p=%.1f: mode 2 %7.1f us  mode 3 (single-bf16 dP) %7.1f us  max diff dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e\n", p, t2b, t4, fq, n1, fk, n2, fv, n3);
      fflush(stdout);
      CK(hipFree(dq2)); CK(hipFree(dk2)); CK(hipFree(dv2));
    }
    if (only_b1) continue;
    // ---- pre-split operand planes (PL kernel variants) against the fp32-input kernels above (variant 1 / dq2)
    {
      const long rows = (long)B * N;
      static void *qp = nullptr, *kp, *vp, *gp;
      if (!qp) { CK(hipMalloc(&qp, rows * 512)); CK(hipMalloc(&kp, rows * 512)); CK(hipMalloc(&vp, rows * 512)); CK(hipMalloc(&gp, rows * 512)); }
      const float ik = focr_attention_keep_scale(p);
      focr_attention_make_planes(q, qp, rows, D, scale * LOG2E, 0);
      focr_attention_make_planes(k, kp, rows, D, 1.f, 0);
      focr_attention_make_planes(v, vp, rows, D, 1.f, 0);
      focr_attention_make_planes(dO, gp, rows, D, ik, 0);
      float tmk = timeit([&]() { focr_attention_make_planes(q, qp, rows, D, scale * LOG2E, 0); });
      g_tune[FOCR_TUNE_ATTN_FWD_VARIANT] = 1;
      focr_attention_fwd(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, p, 1234, 0);
      int rc = focr_attention_planes_fwd(qp, kp, vp, o1, lse1, mask, B, H, N, 256, D, p, 1234, 1, 0);
      if (rc) { printf("planes fwd failed %d\n", rc); return 1; }
      float f0 = timeit([&]() { focr_attention_fwd_premasked(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, p > 0 ? p : 0.1f, 0); }, 8);
      if (p == 0.f) f0 = timeit([&]() { focr_attention_fwd(q, k, v, o0, lse0, mask, B, H, N, D, D, scale, 0.f, 1234, 0); }, 8);
      float f1 = timeit([&]() { focr_attention_planes_fwd(qp, kp, vp, o1, lse1, mask, B, H, N, 256, D, p, 1234, 1, 0); }, 8);
      CK(hipDeviceSynchronize());
      double mxo, eo = maxdiff(o0, o1, n, &mxo), mxl, el = maxdiff(lse0, lse1, (long)B * H * N, &mxl);
      // backward: D from the reference path's prep (work), same keep bits
      g_tune[FOCR_TUNE_ATTN_BWD_DQ_VARIANT] = 1;
      focr_attention_bwd(q, k, v, o0, dO, lse0, mask, dq0, dk0, dv0, work, B, H, N, D, D, scale, p, 0);
      rc = focr_attention_planes_bwd(qp, kp, vp, gp, lse0, work, mask, dq1, dk1, dv1, B, H, N, 256, 256, D, scale, p, 0);
      if (rc) { printf("planes bwd failed %d\n", rc); return 1; }
      float bb0 = timeit([&]() { focr_attention_bwd(q, k, v, nullptr, dO, lse0, mask, dq0, dk0, dv0, work, B, H, N, D, D, scale, p, 0); }, 8);
      float bb1 = timeit([&]() { focr_attention_planes_bwd(qp, kp, vp, gp, lse0, work, mask, dq1, dk1, dv1, B, H, N, 256, 256, D, scale, p, 0); }, 8);
      CK(hipDeviceSynchronize());
      double a1, a2, a3;
      double dq_e = maxdiff(dq0, dq1, n, &a1), dk_e = maxdiff(dk0, dk1, n, &a2), dv_e = maxdiff(dv0, dv1, n, &a3);
      printf("PLANES p=%.1f: fwd fp32-in %7.1f us  planes %7.1f us (make_planes %5.1f us each)  max|dO| %.2e of %.2e  max|dLSE| %.2e\n"
             "              bwd (no prep) fp32-in %7.1f us  planes %7.1f us  max diff dq %.2e/%.2e dk %.2e/%.2e dv %.2e/%.2e\n",
             p, f0, f1, tmk, eo, mxo, el, bb0, bb1, dq_e, a1, dk_e, a2, dv_e, a3);
      fflush(stdout);
      // the same two forwards on the PACKED layouts the training step uses: fp32 [rows][q | k | v] (pitch 384 floats),
      // split rows [rows][Q | K | V] (pitch 768 bf16)
      static float* qkv3 = nullptr; static void* qkvp3 = nullptr;
      if (!qkv3) { CK(hipMalloc(&qkv3, rows * 384 * 4)); CK(hipMalloc(&qkvp3, rows * 768 * 2)); }
      CK(hipMemcpy2D(qkv3, 384 * 4, q, 128 * 4, 128 * 4, rows, hipMemcpyDeviceToDevice));
      CK(hipMemcpy2D(qkv3 + 128, 384 * 4, k, 128 * 4, 128 * 4, rows, hipMemcpyDeviceToDevice));
      CK(hipMemcpy2D(qkv3 + 256, 384 * 4, v, 128 * 4, 128 * 4, rows, hipMemcpyDeviceToDevice));
      CK(hipMemcpy2D(qkvp3, 1536, qp, 512, 512, rows, hipMemcpyDeviceToDevice));
      CK(hipMemcpy2D((char*)qkvp3 + 512, 1536, kp, 512, 512, rows, hipMemcpyDeviceToDevice));
      CK(hipMemcpy2D((char*)qkvp3 + 1024, 1536, vp, 512, 512, rows, hipMemcpyDeviceToDevice));
      const float pp = p > 0 ? p : 0.f;
      auto fpk = [&]() { if (pp > 0) focr_attention_fwd_premasked(qkv3, qkv3 + 128, qkv3 + 256, o0, lse0, mask, B, H, N, 384, D, scale, pp, 0);
                         else focr_attention_fwd(qkv3, qkv3 + 128, qkv3 + 256, o0, lse0, mask, B, H, N, 384, D, scale, 0.f, 1, 0); };
      auto ppk = [&]() { focr_attention_planes_fwd(qkvp3, (char*)qkvp3 + 512, (char*)qkvp3 + 1024, o1, lse1, mask, B, H, N, 768, D, pp, 1, 1, 0); };
      fpk(); ppk();
      float g0 = 1e9f, g1 = 1e9f;
      for (int rep = 0; rep < 3; ++rep) { g0 = std::min(g0, timeit(fpk, 6)); g1 = std::min(g1, timeit(ppk, 6)); }
      CK(hipDeviceSynchronize());
      double mq, eq2 = maxdiff(o0, o1, n, &mq);
      printf("PACKED p=%.1f: fwd fp32 packed %7.1f us  split packed %7.1f us  max|dO| %.2e\n", p, g0, g1, eq2);
      // backward on the packed fp32 layout, dO / O rows at pitch 128 floats vs embedded in a pitch-384 buffer
      static float *dq3 = nullptr, *do3 = nullptr;
      if (!dq3) { CK(hipMalloc(&dq3, rows * 384 * 4)); CK(hipMalloc(&do3, rows * 384 * 4)); }
      CK(hipMemcpy2D(do3, 384 * 4, dO, 128 * 4, 128 * 4, rows, hipMemcpyDeviceToDevice));
      auto b128 = [&]() { focr_attention_bwd(qkv3, qkv3 + 128, qkv3 + 256, nullptr, dO, lse0, mask, dq3, dq3 + 128, dq3 + 256, work, B, H, N, 384, 128, scale, pp, 0); };
      auto b384 = [&]() { focr_attention_bwd(qkv3, qkv3 + 128, qkv3 + 256, nullptr, do3, lse0, mask, dq3, dq3 + 128, dq3 + 256, work, B, H, N, 384, 384, scale, pp, 0); };
      b128(); b384();
      float h0 = 1e9f, h1 = 1e9f;
      for (int rep = 0; rep < 3; ++rep) { h0 = std::min(h0, timeit(b128, 6)); h1 = std::min(h1, timeit(b384, 6)); }
      printf("PACKED p=%.1f: bwd (q k v dq dk dv packed) dO pitch 128: %7.1f us   dO pitch 384: %7.1f us\n", p, h0, h1);
      fflush(stdout);
    }
  }
  return 0;
}

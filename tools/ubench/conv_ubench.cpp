// Standalone micro-benchmark + correctness check of the convolution kernels (no torch: runs in seconds on the GPU box).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/ubench/conv_ubench.cpp -o build/conv_ubench
// For every shape: naive fp32 reference (GPU), generic bf16x3 kernel (conv_bx3.hip), halo kernel (planes 2 and 1).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../fudanocr_amd/csrc/conv3x3_halo.hip"
#include "../../fudanocr_amd/csrc/conv_bx3.hip"

extern "C" void focr_set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); }
extern "C" int focr_get_precision(void) { return 1; }
extern "C" int focr_get_tuning(int) { return 1; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void ref_conv_kernel(const float* X, const float* Wt, const float* bias, const float* R, float* Y, int N, int H,
                                int W, int Cin, int Cout, int KH, int KW, int pad) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)N * H * W * Cout;
  if (i >= total) return;
  int co = i % Cout; long p = i / Cout; int ox = p % W; long t = p / W; int oy = t % H; int n = t / H;
  double s = 0.0;
  for (int kh = 0; kh < KH; ++kh) for (int kw = 0; kw < KW; ++kw) {
    int iy = oy + kh - pad, ix = ox + kw - pad;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const float* xp = X + ((size_t)(n * H + iy) * W + ix) * Cin;
    const float* wp = Wt + ((size_t)(co * KH + kh) * KW + kw) * Cin;
    for (int c = 0; c < Cin; ++c) s += (double)xp[c] * wp[c];
  }
  float v = (float)s + (bias ? bias[co] : 0.f);
  if (R) v += R[p * Cout + co];
  Y[i] = v;
}
__global__ void fill_kernel(float* p, long n, uint32_t seed, float scale) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint32_t h = hash32((uint32_t)i * 2654435761u + seed); p[i] = scale * ((h >> 8) * (1.f / 8388608.f) - 1.f); }
}
static float* dalloc(long n, uint32_t seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), 256, 0, 0, p, n, seed, scale);
  return p;
}
static double maxabs(const std::vector<float>& a) { double m = 0; for (float v : a) m = std::max(m, (double)fabsf(v)); return m; }
static double maxdiff(const std::vector<float>& a, const std::vector<float>& b) { double m = 0; for (size_t i = 0; i < a.size(); ++i) { double d = fabs((double)a[i] - b[i]); if (!(d <= m)) m = d; } return m; }

// median over `iters` samples of (REP back-to-back launches between two events) / REP: the per-launch time inside a
// stream of dependent kernels, as in the training step (a single launch between events adds ~5 us of event overhead)
template <class F> static float timeit(F fn, int iters = 12) {
  const int REP = 10;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) fn();
  std::vector<float> ts;
  for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a, 0)); for (int r = 0; r < REP; ++r) fn(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3f / REP); }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }

struct Shape { const char* name; int N, H, W, Cin, Cout; int flip; };

// host-only layout checks (no GPU): halo swizzle is a bijection and every ds_read_b128 lane group of an A fragment
// read touches 16 distinct 16-byte slots of the 256-byte bank row; weight fragment index is a bijection
static int layout_selftest() {
  std::vector<int> seen(H3_PLANE_BYTES / 16, 0);
  for (int r = 0; r < H3_HR; ++r) for (int p = 0; p < H3_HP; ++p) for (int c = 0; c < 8; ++c) {
    int o = h3_off(r, p, c);
    if (o % 16 || o < 0 || o >= H3_PLANE_BYTES || seen[o / 16]++) { printf("h3_off collision at %d %d %d\n", r, p, c); return 1; }
  }
  const int groups[4][16] = {{0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27}, {4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31},
                             {32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59}, {36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63}};
  for (int wave = 0; wave < 4; ++wave) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) for (int kk = 0; kk < 4; ++kk)
    for (int gsel = 0; gsel < 4; ++gsel) {
      int used[16] = {0};
      for (int i = 0; i < 16; ++i) {
        int lane = groups[gsel][i], li = lane & 31, lh = lane >> 5;
        int a = h3_off(wave + kh, li + kw, 2 * kk + lh);
        int slot = (a / 16) % 16;
        if (used[slot]++) { printf("bank conflict wave %d kh %d kw %d kk %d group %d\n", wave, kh, kw, kk, gsel); return 1; }
      }
    }
  const int rows = 96, K = 80, ks = (K + 15) / 16;
  std::vector<int> s2(wprep_elems(rows, K), 0);
  for (int pl = 0; pl < 2; ++pl) for (int n = 0; n < rows; ++n) for (int k = 0; k < K; ++k) {
    size_t i = wfrag_index(n, k, ks, pl);
    if (i >= s2.size() || s2[i]++) { printf("wfrag collision\n"); return 1; }
  }
  printf("layout selftest OK\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "layout")) return layout_selftest();
  int B = argc > 1 ? atoi(argv[1]) : 128;
  std::vector<Shape> shapes = {
      {"srb 3x3 64->64 16x64", B, 16, 64, 64, 64, 0},
      {"srb dgrad 64->64 (flipped w)", B, 16, 64, 64, 64, 1},
      {"up 3x3 64->256 16x64", B, 16, 64, 64, 256, 0},
      {"up dgrad 256->64 16x64", B, 16, 64, 256, 64, 1},
      {"crnn1 64->128 16x50", B, 16, 50, 64, 128, 0},
      {"crnn2 128->256 8x25", B, 8, 25, 128, 256, 0},
      {"crnn3 256->256 8x25", B, 8, 25, 256, 256, 0},
      {"crnn4 256->512 4x26", B, 4, 26, 256, 512, 0},
      {"crnn5 512->512 4x26", B, 4, 26, 512, 512, 0},
      {"odd 5x37 64->64 (N=3)", 3, 5, 37, 64, 64, 0},
  };
  WPrepDesc* ddesc; CK(hipMalloc(&ddesc, sizeof(WPrepDesc)));
  printf("empty kernel (1024 blocks), back-to-back: %.2f us per launch\n", timeit([&]() { hipLaunchKernelGGL(empty_kernel, dim3(1024), 256, 0, 0, (float*)nullptr); }));
  const char* filt = argc > 2 ? argv[2] : nullptr;
  for (const Shape& s : shapes) {
    if (filt && !strstr(s.name, filt)) continue;
    const long nx = (long)s.N * s.H * s.W * s.Cin, ny = (long)s.N * s.H * s.W * s.Cout, nw = (long)s.Cout * 9 * s.Cin;
    float* x = dalloc(nx, 1, 1.f);
    float* w = dalloc(nw, 2, 0.05f);        // forward-layout weights of the layer computed here
    float* bias = dalloc(s.Cout, 3, 0.5f);
    float* res = dalloc(ny, 4, 1.f);
    if (getenv("H3_UB_NORES")) res = nullptr;          // launches without a residual operand (round 6: their own instantiation)
    float *yref, *y0, *y1, *y2, *stats;
    CK(hipMalloc(&yref, ny * 4)); CK(hipMalloc(&y0, ny * 4)); CK(hipMalloc(&y1, ny * 4)); CK(hipMalloc(&y2, ny * 4));
    const int tiles = focr_conv3x3_halo_tiles(s.N, s.H, s.W);
    CK(hipMalloc(&stats, (size_t)tiles * s.Cout * 2 * 4));
    // flip = 1: the tested layer's weights come from a "parent" layer [Cin_parent = Cout_here][..]: build the parent's
    // OHWI tensor so that flip-prep of it equals forward-prep of w.  parent[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]
    float* wparent = nullptr;
    if (s.flip) {
      std::vector<float> hw(nw), hp(nw);
      CK(hipMemcpy(hw.data(), w, nw * 4, hipMemcpyDeviceToHost));
      for (int co = 0; co < s.Cout; ++co) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) for (int ci = 0; ci < s.Cin; ++ci)
        hp[((size_t)(ci * 3 + (2 - kh)) * 3 + (2 - kw)) * s.Cout + co] = hw[((size_t)(co * 3 + kh) * 3 + kw) * s.Cin + ci];
      CK(hipMalloc(&wparent, nw * 4));
      CK(hipMemcpy(wparent, hp.data(), nw * 4, hipMemcpyHostToDevice));
    }
    __bf16* wf; CK(hipMalloc(&wf, wprep_elems(s.Cout, 9 * s.Cin) * 2));
    WPrepDesc d;
    if (s.flip) d = WPrepDesc{wparent, wf, s.Cin, 3, 3, s.Cout, 1, 0};   // parent: cout = Cin_here, cin = Cout_here
    else d = WPrepDesc{w, wf, s.Cout, 3, 3, s.Cin, 0, 0};
    CK(hipMemcpy(ddesc, &d, sizeof(d), hipMemcpyHostToDevice));
    focr_weight_prep_frag_launch(ddesc, 1, (long)s.Cout * 9 * s.Cin / 8, 0);
    hipLaunchKernelGGL(ref_conv_kernel, dim3((ny + 255) / 256), 256, 0, 0, x, w, bias, res, yref, s.N, s.H, s.W, s.Cin, s.Cout, 3, 3, 1);
    CK(hipDeviceSynchronize());
    const int M = s.N * s.H * s.W;
    auto old_k = [&]() { focr_conv_fwd_bx3(x, w, bias, res, y0, s.N, s.H, s.W, s.Cin, s.H, s.W, s.Cout, 3, 3, 1, 1, M, s.Cout, s.Cout, s.Cin, 1.f, 0, nullptr, 0, 0); };
    auto h2 = [&]() { focr_conv3x3_halo(x, wf, bias, res, y1, nullptr, s.N, s.H, s.W, s.Cin, s.Cout, s.Cin, s.Cout, s.Cout, 1.f, 0, 2, 0); };
    auto h2s = [&]() { focr_conv3x3_halo(x, wf, bias, res, y1, stats, s.N, s.H, s.W, s.Cin, s.Cout, s.Cin, s.Cout, s.Cout, 1.f, 0, 2, 0); };
    auto h1 = [&]() { focr_conv3x3_halo(x, wf, bias, res, y2, nullptr, s.N, s.H, s.W, s.Cin, s.Cout, s.Cin, s.Cout, s.Cout, 1.f, 0, 1, 0); };
    auto prep = [&]() { focr_weight_prep_frag_launch(ddesc, 1, (long)s.Cout * 9 * s.Cin / 8, 0); };
#ifdef H3_TRACE
    if (getenv("H3_TRACE_RUN")) {
      // block timeline of ONE launch (or of the 2nd of 3 back-to-back launches with H3_TRACE_RUN=3)
      unsigned long long* tb; CK(hipMalloc(&tb, (size_t)tiles * 64));
      for (int planes = 2; planes >= 1; --planes) {
        auto fn = [&]() { if (planes == 2) h2(); else h1(); };
        fn(); fn(); CK(hipDeviceSynchronize());
        const bool chain = atoi(getenv("H3_TRACE_RUN")) == 3;
        unsigned long long* nul = nullptr;
        if (chain) fn();
        CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(h3_trace_buf), &tb, sizeof(tb), 0, hipMemcpyHostToDevice, 0));
        fn();
        CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(h3_trace_buf), &nul, sizeof(nul), 0, hipMemcpyHostToDevice, 0));
        if (chain) fn();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)tiles * 8);
        CK(hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t3 = 0;
        for (int i = 0; i < tiles; ++i) { t0 = std::min(t0, h[i * 8]); t3 = std::max(t3, h[i * 8 + 3]); }
        printf("TRACE planes %d tiles %d span %.2f us\n", planes, tiles, (t3 - t0) * 0.01);
        for (int i = 0; i < tiles; ++i) {
          const unsigned hw = (unsigned)h[i * 8 + 4], xcc = (unsigned)h[i * 8 + 5] & 15;
          printf("B %d xcc %u se %u cu %u simd %u wave %u  %.2f %.2f %.2f %.2f\n", i, xcc, (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3,
                 hw & 15, (h[i * 8] - t0) * 0.01, (h[i * 8 + 1] - t0) * 0.01, (h[i * 8 + 2] - t0) * 0.01, (h[i * 8 + 3] - t0) * 0.01);
        }
      }
      return 0;
    }
#endif
    float t_old = timeit(old_k), t_h2 = timeit(h2), t_h1 = timeit(h1), t_h2s = timeit(h2s), t_prep = timeit(prep);
    CK(hipDeviceSynchronize());
    std::vector<float> r(ny), a(ny), b(ny), c(ny);
    CK(hipMemcpy(r.data(), yref, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), y0, ny * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), y1, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), y2, ny * 4, hipMemcpyDeviceToHost));
    // statistics check (after h2s ran last on y1): column sums of y1 vs folded stats
    std::vector<float> st((size_t)tiles * s.Cout * 2);
    CK(hipMemcpy(st.data(), stats, st.size() * 4, hipMemcpyDeviceToHost));
    double serr = 0, smax = 0;
    for (int co = 0; co < s.Cout; co += 7) {
      double s1 = 0, s2 = 0, r1 = 0, r2 = 0;
      for (int t = 0; t < tiles; ++t) { s1 += st[((size_t)t * s.Cout + co) * 2]; s2 += st[((size_t)t * s.Cout + co) * 2 + 1]; }
      for (long p = 0; p < (long)M; ++p) { double v = b[p * s.Cout + co]; r1 += v; r2 += v * v; }
      serr = std::max(serr, std::max(fabs(s1 - r1), fabs(s2 - r2) / (1 + fabs(r2)) * (1 + fabs(r1))));
      smax = std::max(smax, fabs(r1));
    }
    const double mx = maxabs(r);
    const double fl = 2.0 * M * s.Cout * 9.0 * s.Cin;
    printf("%-30s ref|max| %.3g  old %7.1f us (%5.0f TF, err %.1e) | halo x3 %7.1f us (%5.0f TF, err %.1e; +stats %7.1f us, stat err %.1e of %.3g) | halo x1 %7.1f us (err %.1e) | prep %5.1f us\n",
           s.name, mx, t_old, fl / t_old / 1e6, maxdiff(r, a) / mx, t_h2, fl / t_h2 / 1e6, maxdiff(r, b) / mx, t_h2s, serr, smax, t_h1, maxdiff(r, c) / mx, t_prep);
    fflush(stdout);
    hipFree(x); hipFree(w); hipFree(bias); hipFree(res); hipFree(yref); hipFree(y0); hipFree(y1); hipFree(y2); hipFree(stats); hipFree(wf);
    if (wparent) hipFree(wparent);
  }
  return 0;
}

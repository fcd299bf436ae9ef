// Weight gradient of a 3x3 / pad 1 convolution with a handful of input channels (Cin <= 4) and 32 output channels: the
// first layer of the STN head (stn_head.py:34, Conv2d(in_planes, 32, 3, 1, 1)).  The generic fp32 kernel spends 57 us on
// this layer (27 of its 64 K columns used, a memset and atomics) -- and it is the LAST weight gradient of the step: the
// optimizer waits for it (profiles/r04_step_sequence.txt).  Here: plain fp32 VALU arithmetic (9 * Cin * 32 outputs, 2 flop
// per pixel and output), per-block partial sums and a fixed-order fold: deterministic, no atomics, no memset.
//   block = a range of image rows; thread = (output channel co, tap group jg); the three input rows around the current
//   row (zero halo) and the dY row live in LDS; per pixel a thread reads ONE dY value and its <= 4 taps' inputs
//   (addresses shared by the 32 threads of a tap group: broadcasts).
#include "focr_common.h"

#define CS_CO 32
#define CS_MAXW 128
#define CS_JPT 5                        // taps per thread: ceil(9 * 4 / 8)

__global__ __launch_bounds__(256) void conv3x3_cin_small_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                                      float* __restrict__ PART, int N, int H, int W, int Cin,
                                                                      int ldd, int rows_per_block) {
  __shared__ float xs[3][(CS_MAXW + 2) * 4];
  __shared__ float ds[CS_MAXW * CS_CO];
  const int tid = threadIdx.x, co = tid & 31, jg = tid >> 5;
  const int K = 9 * Cin;                                   // taps j = (kh * 3 + kw) * Cin + ci
  const int R0 = blockIdx.x * rows_per_block, R1 = min(N * H, R0 + rows_per_block);
  float acc[CS_JPT], bacc = 0.f;
  int xoff[CS_JPT];                                        // LDS offset of tap j relative to pixel px: row kh, column px + kw
#pragma unroll
  for (int u = 0; u < CS_JPT; ++u) {
    acc[u] = 0.f;
    const int j = jg + 8 * u;
    const int t = j / Cin, ci = j - t * Cin, kh = t / 3, kw = t - kh * 3;
    xoff[u] = j < K ? kh * ((CS_MAXW + 2) * 4) + kw * Cin + ci : 0;      // (taps past K: a valid address, result not stored)
  }
  for (int g = R0; g < R1; ++g) {
    const int iy = g % H;
    __syncthreads();                                       // the previous row's reads are over
    for (int i = tid; i < 3 * (W + 2) * Cin; i += 256) {
      const int r = i / ((W + 2) * Cin), c = i - r * ((W + 2) * Cin);
      const int px = c / Cin - 1, ci = c - (px + 1) * Cin, yy = iy + r - 1;
      xs[r][c] = ((unsigned)yy < (unsigned)H && (unsigned)px < (unsigned)W)
                     ? X[((size_t)(g + r - 1) * W + px) * Cin + ci] : 0.f;
    }
    for (int i = tid; i < W * (CS_CO / 4); i += 256) {
      const int px = i >> 3, c4 = (i & 7) * 4;
      *reinterpret_cast<float4*>(&ds[px * CS_CO + c4]) = *reinterpret_cast<const float4*>(dY + ((size_t)g * W + px) * ldd + c4);
    }
    __syncthreads();
    const float* xb = &xs[0][0];
    // eight pixels per iteration: 8 + 40 LDS reads in flight before the 40 fma (one pixel at a time the loop waited out the
    // LDS latency 64 times per row: 55 us for the layer)
    for (int px0 = 0; px0 < W; px0 += 8) {
      float d[8], xv[8][CS_JPT];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int px = min(px0 + e, W - 1);
        d[e] = px0 + e < W ? ds[px * CS_CO + co] : 0.f;
#pragma unroll
        for (int u = 0; u < CS_JPT; ++u) xv[e][u] = xb[xoff[u] + px * Cin];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bacc += d[e];
#pragma unroll
        for (int u = 0; u < CS_JPT; ++u) acc[u] = fmaf(d[e], xv[e][u], acc[u]);
      }
    }
  }
  // partial tile of this block: [co][j] (the dW layout [Cout][3][3][Cin]) + the bias column
  float* part = PART + (size_t)blockIdx.x * (CS_CO * (K + 1));
#pragma unroll
  for (int u = 0; u < CS_JPT; ++u) {
    const int j = jg + 8 * u;
    if (j < K) part[co * K + j] = acc[u];
  }
  if (jg == 0) part[CS_CO * K + co] = bacc;
}

// dw[i] (and dbias) = sum over the blocks in a fixed order: 32 lanes per output element (lane l adds blocks l, l + 32, ...),
// then a butterfly over the 32 lanes
__global__ __launch_bounds__(256) void conv3x3_cin_small_fold_kernel(const float* __restrict__ PART, float* __restrict__ dw,
                                                                     float* __restrict__ dbias, int nb, int K) {
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31, per = CS_CO * (K + 1);
  float s = 0.f;
  if (i < per) {
    for (int b = l; b < nb; b += 32 * 4) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = b + 32 * u < nb ? PART[(size_t)(b + 32 * u) * per + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u];
    }
  }
#pragma unroll
  for (int w = 16; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if (i < per && l == 0) {
    if (i < CS_CO * K) dw[i] = s;
    else if (dbias) dbias[i - CS_CO * K] = s;
  }
}

static bool cs_applicable(int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx) {
  return KH == 3 && KW == 3 && padH == 1 && padW == 1 && Cin >= 1 && Cin <= 4 && Cout == CS_CO && W <= CS_MAXW && ldx == Cin &&
         ldd % 4 == 0 && H >= 1;
}
static void cs_blocks(int N, int H, int& nb, int& rpb) {
  const int rows = N * H;
  nb = rows < 512 ? rows : 512;
  rpb = cdiv(rows, nb);
  nb = cdiv(rows, rpb);
}
long focr_conv3x3_cin_small_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW) {
  if (!cs_applicable(H, W, Cin, Cout, KH, KW, padH, padW, 4, Cin)) return 0;
  int nb, rpb;
  cs_blocks(N, H, nb, rpb);
  return (long)nb * CS_CO * (9 * Cin + 1);
}
// launcher used by focr_conv2d_wgrad (conv_igemm.hip); returns 1 if the layer was handled here.  dw / dbias are OVERWRITTEN.
int focr_conv3x3_cin_small_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                                 int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx,
                                 hipStream_t stream) {
  if (!cs_applicable(H, W, Cin, Cout, KH, KW, padH, padW, ldd, ldx)) return 0;
  int nb, rpb;
  cs_blocks(N, H, nb, rpb);
  const int K = 9 * Cin;
  if (!ws || ws_floats < (long)nb * CS_CO * (K + 1)) return 0;
  hipLaunchKernelGGL(conv3x3_cin_small_wgrad_kernel, dim3(nb), 256, 0, stream, x, dy, ws, N, H, W, Cin, ldd, rpb);
  hipLaunchKernelGGL(conv3x3_cin_small_fold_kernel, dim3(cdiv(CS_CO * (K + 1), 8)), 256, 0, stream, (const float*)ws, dw,
                     dbias, nb, K);
  return 1;
}

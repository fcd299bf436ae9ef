// 9x9, pad 4 convolution with THREE input channels and Cout % 32 == 0 (block1 of TSRN/TBSRN, tsrn.py:28-31 /
// tbsrn.py:178-181, and the data gradient of the 9x9 output layer block8.1), bf16x3.
//
// The generic kernel gathers the 243 (kh, kw, c) inputs of every pixel with scalar loads.  In NHWC the 27 values
// (kw, c) of one tap row are CONTIGUOUS: X[n, oy+kh-4, ox-4 .. ox+4, 0..2].  So K is laid out as 9 segments of 32
// (27 + 5 zero-weight pads) = 18 MFMA k-steps:
//   * a block owns one 32-channel output slice and a row range of one image; its weight fragments (9 x 2 k-steps,
//     hi/lo) are split once and stay in REGISTERS of every wave (144 VGPRs);
//   * input rows live in LDS as plain fp32 with a 4-pixel zero halo, ten-slot rolling window (one new row per output
//     row, staged while the current row is multiplied: one barrier per row);
//   * wave = 32 pixels x 32 channels; an A fragment is 8 consecutive floats of a staged row starting at
//     3*(px) + 16*s + 8*lh (lane stride 3 floats: conflict-free ds_read_b32), split to bf16 hi/lo on the fly.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 cbf16x8;

#define C3_SLOTS 10
#define C3_PAD 16          // floats of slack after each staged row (fragment reads of the pad k-slots run past the row)

template <int NTILE>       // 32-pixel tiles per row = waves per block (W = 32 * NTILE)
__global__ __launch_bounds__(64 * NTILE) void conv9x9_cin3_bx3_kernel(const float* __restrict__ X,
                                                                      const float* __restrict__ Wt,   // [Cout][9][9][3]
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ Y, int H, int W, int Cout,
                                                                      int ldy, int RR, int Rrows, float alpha) {
  extern __shared__ __attribute__((aligned(16))) float rows_c3[];     // [C3_SLOTS][(W + 8) * 3 + C3_PAD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rowlen = (W + 8) * 3 + C3_PAD;
  const int co0 = blockIdx.y * 32;
  const int n = blockIdx.x / RR, rr = blockIdx.x - n * RR;
  const int r0 = rr * Rrows, r1 = min(H, r0 + Rrows);
  if (r0 >= r1) return;
  // weight fragments: B[co = co0 + li][k = kh*32 + 16 s + 8 lh + e], k-slot j = 16 s + 8 lh + e < 27 -> (kw, c) = j
  cbf16x8 bh[9][2], bl[9][2];
#pragma unroll
  for (int kh = 0; kh < 9; ++kh)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 16 * s + 8 * lh + e;
        const float v = j < 27 ? Wt[((size_t)(co0 + li) * 9 + kh) * 27 + j] : 0.f;
        const __bf16 h = (__bf16)v;
        bh[kh][s][e] = h;
        bl[kh][s][e] = (__bf16)(v - (float)h);
      }
    }
  const float bco = bias ? bias[co0 + li] : 0.f;
  const float* ximg = X + (size_t)n * H * W * 3;
  auto stage_row = [&](int iy) {                       // all threads: one input row (or zeros) into its slot
    float* dst = rows_c3 + ((iy + C3_SLOTS * 4) % C3_SLOTS) * rowlen;
    const bool ok = (unsigned)iy < (unsigned)H;
    const float* src = ximg + (size_t)(ok ? iy : 0) * W * 3;
    for (int i = tid; i < rowlen; i += 64 * NTILE) {
      const int j = i - 12;                            // 4 halo pixels x 3 channels on the left
      dst[i] = (ok && j >= 0 && j < W * 3) ? src[j] : 0.f;
    }
  };
  for (int iy = r0 - 4; iy < r0 + 4; ++iy) stage_row(iy);
  for (int oy = r0; oy < r1; ++oy) {
    stage_row(oy + 4);     // its slot held row oy - 6, last read two iterations ago; one barrier per row bounds the skew
    __syncthreads();       // between waves to less than one iteration, so no second barrier is needed
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 9; ++kh) {
      const float* rowp = rows_c3 + ((oy + kh - 4 + C3_SLOTS * 4) % C3_SLOTS) * rowlen + (wave * 32 + li) * 3 + 8 * lh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        cbf16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = rowp[16 * s + e];
          const __bf16 h = (__bf16)v;
          ah[e] = h;
          al[e] = (__bf16)(v - (float)h);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kh][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kh][s], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kh][s], acc, 0, 0, 0);
      }
    }
    float* yrow = Y + ((size_t)(n * H + oy) * W + wave * 32) * ldy + co0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = (r & 3) + 8 * (r >> 2) + 4 * lh;
      yrow[(size_t)px * ldy] = alpha * acc[r] + bco;
    }
  }
}

int focr_conv9x9_cin4_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                          int ldy, float alpha, hipStream_t stream);

// used by focr_conv2d_fwd (conv_igemm.hip); returns 1 if the layer was handled here
int focr_conv9x9_cin3_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                          int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldx, int ldy,
                          float alpha, int relu, hipStream_t stream) {
  if (KH == 9 && KW == 9 && padH == 4 && padW == 4 && Cin == 4 && ldx == 4 && !residual && !relu)   // --mask: conv9x9_cin4.hip
    return focr_conv9x9_cin4_fwd(x, w, bias, y, N, H, W, Cout, ldy, alpha, stream);
  if (!(KH == 9 && KW == 9 && padH == 4 && padW == 4 && Cin == 3 && ldx == 3 && Cout % 32 == 0 && !residual && !relu &&
        (W == 64 || W == 128)))
    return 0;
  int RR = cdiv(512, N * (Cout / 32));                 // >= 2 blocks per CU
  if (RR > H / 4) RR = H / 4;
  if (RR < 1) RR = 1;
  const int Rrows = cdiv(H, RR);
  RR = cdiv(H, Rrows);
  const size_t lds = (size_t)C3_SLOTS * ((W + 8) * 3 + C3_PAD) * sizeof(float);
  dim3 grid(N * RR, Cout / 32);
  if (W == 128)
    hipLaunchKernelGGL((conv9x9_cin3_bx3_kernel<4>), grid, 256, lds, stream, x, w, bias, y, H, W, Cout, ldy, RR, Rrows, alpha);
  else
    hipLaunchKernelGGL((conv9x9_cin3_bx3_kernel<2>), grid, 128, lds, stream, x, w, bias, y, H, W, Cout, ldy, RR, Rrows, alpha);
  return 1;
}

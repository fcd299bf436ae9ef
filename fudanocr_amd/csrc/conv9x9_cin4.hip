// 9x9, pad 4 convolution with FOUR input channels and Cout % 32 == 0: block1 of TSRN / TBSRN under the reference's
// `--mask` (main.py:31: in_planes = 4, tsrn.py:24-31 / tbsrn.py:174-181) and, on flipped weights, the data gradient of
// the 9x9 output layer block8.1 (64 -> 4).  Split-bf16 products, fp32 accumulate (precision modes 1-3).
//
// conv9x9_cin3.hip's layout does not carry over: with three channels a tap row's 27 (kw, c) values are contiguous floats
// read at a lane stride of 3 words (odd: conflict-free); with four the stride is 4 words (a 4-way bank conflict for
// ds_read_b32) and a 36-value segment needs 3 k-steps per tap row (25 % padding).  But four channels make every TAP one
// aligned 16-byte vector, so here
//   * K is the flat tap list: k-slot (step s, lane half lh, e) <-> tap t = 4 s + 2 lh + (e >> 2), channel e & 3;
//     81 taps + 3 zero-weight pads = 21 k-steps (3.6 % padding), 63 MFMAs per 32 pixel x 32 channel tile;
//   * an A fragment is TWO ds_read_b128 (taps t, t + 1 of pixel px: LDS float4 index row(kh) + px + kw), lanes = 32
//     consecutive pixels = 512 contiguous bytes: conflict-free; split to bf16 hi / lo in registers;
//   * input rows live in LDS as float4 per pixel with a 4-pixel zero halo, ten-slot rolling window, one new row staged
//     per output row, ONE barrier per row (as in the three-channel kernel);
//   * the block's 32-channel weight slice is split ONCE into MFMA-fragment order in LDS (21 x 2 planes x 1 KB): a B
//     fragment is one lane-linear ds_read_b128 per plane (144 VGPRs of register-resident fragments at three channels
//     would be 168 here);
//   * pad taps (t > 80) have zero weights; their A reads are redirected to tap 80's address so that no uninitialised LDS
//     (a NaN pattern times zero) enters the products.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 c4bf16x8;

#define C4_SLOTS 10
#define C4_KSTEPS 21

template <int NTILE>       // 32-pixel tiles per row = waves per block (W = 32 * NTILE)
__global__ __launch_bounds__(64 * NTILE) void conv9x9_cin4_bx3_kernel(const float* __restrict__ X,
                                                                      const float* __restrict__ Wt,   // [Cout][9][9][4]
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ Y, int H, int Cout, int ldy,
                                                                      int RR, int Rrows, float alpha) {
  constexpr int W = 32 * NTILE, RL = W + 8;                 // staged row length in pixels (float4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c4[];
  float4* const rows = reinterpret_cast<float4*>(smem_c4);                                  // [C4_SLOTS][RL]
  c4bf16x8* const wf = reinterpret_cast<c4bf16x8*>(smem_c4 + C4_SLOTS * RL * 16);           // [21][2 planes][64 lanes]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int co0 = blockIdx.y * 32;
  const int n = blockIdx.x / RR, rr = blockIdx.x - n * RR;
  const int r0 = rr * Rrows, r1 = min(H, r0 + Rrows);
  if (r0 >= r1) return;
  // ---- weight fragments: lane (li_, lh_) of k-step s holds taps 4 s + 2 lh_ (+1) of output channel co0 + li_
  for (int i = tid; i < C4_KSTEPS * 64; i += 64 * NTILE) {
    const int s = i >> 6, l = i & 63, t0 = 4 * s + 2 * (l >> 5);
    const float* wp = Wt + ((size_t)(co0 + (l & 31)) * 81) * 4;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = t0 < 81 ? *reinterpret_cast<const float4*>(wp + 4 * t0) : z;
    const float4 b = t0 + 1 < 81 ? *reinterpret_cast<const float4*>(wp + 4 * (t0 + 1)) : z;
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    c4bf16x8 hi, lo;
    focr_split8(v, hi, lo);
    wf[(s * 2 + 0) * 64 + l] = hi;
    wf[(s * 2 + 1) * 64 + l] = lo;
  }
  const float bco = bias ? bias[co0 + li] : 0.f;
  const float4* ximg = reinterpret_cast<const float4*>(X) + (size_t)n * H * W;
  auto stage_row = [&](int iy) {                       // all threads: one input row (or zeros) into its slot
    float4* dst = rows + ((iy + C4_SLOTS * 4) % C4_SLOTS) * RL;
    const bool ok = (unsigned)iy < (unsigned)H;
    const float4* src = ximg + (size_t)(ok ? iy : 0) * W;
    for (int i = tid; i < RL; i += 64 * NTILE) {
      const int j = i - 4;                             // 4 halo pixels on the left
      dst[i] = (ok && j >= 0 && j < W) ? src[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  for (int iy = r0 - 4; iy < r0 + 4; ++iy) stage_row(iy);
  const int px = wave * 32 + li;
  for (int oy = r0; oy < r1; ++oy) {
    stage_row(oy + 4);     // its slot held row oy - 6, last read two iterations ago; one barrier per row bounds the skew
    __syncthreads();       // between waves to less than one iteration (and publishes the weight fragments the first time)
    int sb[9];             // float4 index of the slot of input row oy + kh - 4 (block-uniform)
#pragma unroll
    for (int kh = 0; kh < 9; ++kh) sb[kh] = ((oy + kh - 4 + C4_SLOTS * 4) % C4_SLOTS) * RL;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < C4_KSTEPS; ++s) {
      // taps of the two lane halves (compile-time), pads redirected to tap 80
      constexpr int TMAX = 80;
      const int ta0 = 4 * s < TMAX ? 4 * s : TMAX, ta1 = 4 * s + 1 < TMAX ? 4 * s + 1 : TMAX;
      const int tb0 = 4 * s + 2 < TMAX ? 4 * s + 2 : TMAX, tb1 = 4 * s + 3 < TMAX ? 4 * s + 3 : TMAX;
      const int oa0 = sb[ta0 / 9] + ta0 % 9, oa1 = sb[ta1 / 9] + ta1 % 9;
      const int ob0 = sb[tb0 / 9] + tb0 % 9, ob1 = sb[tb1 / 9] + tb1 % 9;
      const float4 a = rows[(lh ? ob0 : oa0) + px];
      const float4 b = rows[(lh ? ob1 : oa1) + px];
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      c4bf16x8 ah, al;
      focr_split8(v, ah, al);
      const c4bf16x8 bh = wf[(s * 2 + 0) * 64 + lane], bl = wf[(s * 2 + 1) * 64 + lane];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
    float* yrow = Y + ((size_t)(n * H + oy) * W + wave * 32) * ldy + co0 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;
      yrow[(size_t)p * ldy] = alpha * acc[r] + bco;
    }
  }
}

// used by focr_conv9x9_cin3_fwd (conv9x9_cin3.hip); returns 1 if the layer was handled here
int focr_conv9x9_cin4_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cout,
                          int ldy, float alpha, hipStream_t stream) {
  if (!(Cout % 32 == 0 && (W == 64 || W == 128) && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(w) & 15) == 0))
    return 0;
  int RR = cdiv(512, N * (Cout / 32));                 // >= 2 blocks per CU
  if (RR > H / 4) RR = H / 4;
  if (RR < 1) RR = 1;
  const int Rrows = cdiv(H, RR);
  RR = cdiv(H, Rrows);
  const int lds = C4_SLOTS * (W + 8) * 16 + C4_KSTEPS * 2 * 1024;       // 64 768 B at W = 128: two blocks per CU
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv9x9_cin4_bx3_kernel<4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C4_SLOTS * 136 * 16 + C4_KSTEPS * 2048) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv9x9_cin4_bx3_kernel<2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C4_SLOTS * 72 * 16 + C4_KSTEPS * 2048) != hipSuccess)
      return 0;
    focr_dev_mark(attr_set);
  }
  dim3 grid(N * RR, Cout / 32);
  if (W == 128)
    hipLaunchKernelGGL((conv9x9_cin4_bx3_kernel<4>), grid, 256, lds, stream, x, w, bias, y, H, Cout, ldy, RR, Rrows, alpha);
  else
    hipLaunchKernelGGL((conv9x9_cin4_bx3_kernel<2>), grid, 128, lds, stream, x, w, bias, y, H, Cout, ldy, RR, Rrows, alpha);
  return 1;
}

// SR output layer: Conv2d(64 -> Cout<=3, 9x9, pad 4) (model/tsrn.py:43, tbsrn.py:197), fp32 MFMA.
//
// A 3-channel output wastes >90 % of a 32-wide MFMA N tile in the generic implicit GEMM
// (profiles/r01: 1.77 ms forward + 7.5 ms wgrad per step at B=128).  Here the 9 horizontal
// taps are folded into N:  j = co*9 + kw  (27 of 32 columns useful):
//
//   forward, per output row (n,oy) and per kh:
//       Z[ix][j] = sum_ci X[n,oy+kh-4,ix,ci] * W[co][kh][kw][ci]          (MFMA, K = 64)
//       y[ox][co] += sum_kw Z[ox+kw-4][co*9+kw]                            (diagonal gather in LDS)
//   wgrad, per input row (n,iy), for kh in a group of 3:
//       dW[co][kh][kw][ci] += sum_ix X[n,iy,ix,ci] * dY[n,iy-kh+4,ix-kw+4,co]
//       = MFMA with A = X^T (ci x pixels), B[pixel][j] gathered from the dY row in LDS.
//
// Executed/useful flop ratio 32/27 instead of 32/3.
#include "focr_common.h"

#define C9 64          // input channels (fixed)
#define XP 68          // LDS pitch of pixel rows read with ds_read_b128 fragments
#define MAXT 5         // up to 5 tiles of 32 input pixels  (W + 8 <= 160)

__global__ __launch_bounds__(320) void conv9x9_out_fwd_kernel(const float* __restrict__ X,
                                                              const float* __restrict__ Wt,   // [Cout][9][9][64]
                                                              const float* __restrict__ bias,
                                                              float* __restrict__ Y, int H, int W, int Cout) {
  __shared__ __attribute__((aligned(16))) float Xs[MAXT * 32 * XP];
  __shared__ __attribute__((aligned(16))) float Ws[32 * XP];
  __shared__ float Zs[MAXT * 32][33];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x / H, oy = blockIdx.x % H;
  const int ntile = (W + 8 + 31) / 32;
  // zero the padding rows 27..31 of the weight tile once
  for (int i = tid; i < 32 * XP; i += 320) Ws[i] = 0.f;
  // each thread owns up to 2 outputs (ox,co): o = tid, tid+320
  float yacc[2] = {0.f, 0.f};
  const int nout = W * Cout;
  __syncthreads();
  for (int kh = 0; kh < 9; ++kh) {
    const int iy = oy + kh - 4;
    if ((unsigned)iy >= (unsigned)H) continue;           // uniform across the block
    // ---- stage the input row (with 4-pixel halos) and the 27 weight rows of this kh ----
    const float* xrow = X + ((size_t)n * H + iy) * W * C9;
    for (int i = tid; i < ntile * 32 * 16; i += 320) {
      int px = i >> 4, c4 = i & 15;
      int ix = px - 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((unsigned)ix < (unsigned)W) v = *reinterpret_cast<const float4*>(xrow + (size_t)ix * C9 + c4 * 4);
      *reinterpret_cast<float4*>(&Xs[px * XP + c4 * 4]) = v;
    }
    for (int i = tid; i < Cout * 9 * 16; i += 320) {
      int j = i >> 4, c4 = i & 15;
      int co = j / 9, kw = j - co * 9;
      *reinterpret_cast<float4*>(&Ws[j * XP + c4 * 4]) =
          *reinterpret_cast<const float4*>(Wt + (((size_t)co * 9 + kh) * 9 + kw) * C9 + c4 * 4);
    }
    __syncthreads();
    if (wave < ntile) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ap = &Xs[(wave * 32 + li) * XP + 4 * lh];
      const float* bp = &Ws[li * XP + 4 * lh];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float4 a = *reinterpret_cast<const float4*>(ap + 8 * t);
        float4 b = *reinterpret_cast<const float4*>(bp + 8 * t);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Zs[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int o = tid + 320 * e;
      if (o < nout) {
        int ox = o / Cout, co = o - ox * Cout;
        float s = 0.f;
#pragma unroll
        for (int kw = 0; kw < 9; ++kw) s += Zs[ox + kw][co * 9 + kw];     // Zs row = ix + 4
        yacc[e] += s;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    int o = tid + 320 * e;
    if (o < nout) {
      int co = o % Cout;
      Y[((size_t)n * H + oy) * W * Cout + o] = yacc[e] + (bias ? bias[co] : 0.f);
    }
  }
}

// grid (3 kh-groups, chunks); block 256 = 4 waves, wave w owns pixels [32w, 32w+32) of the row (W<=128)
__global__ __launch_bounds__(256) void conv9x9_out_wgrad_kernel(const float* __restrict__ X,
                                                                const float* __restrict__ dY,
                                                                float* __restrict__ dW, int N, int H, int W,
                                                                int Cout, int rows_per_chunk) {
  __shared__ __attribute__((aligned(16))) float Xs[128 * C9];
  __shared__ float Gs[3][128 * 3 + 4];
  __shared__ float Red[4][32][33];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int khg = blockIdx.x;                               // kh in {3*khg, 3*khg+1, 3*khg+2}
  const int row0 = blockIdx.y * rows_per_chunk;
  const int row1 = min(N * H, row0 + rows_per_chunk);
  const int jco = li / 9, jkw = li - jco * 9;
  const bool jok = li < Cout * 9;
  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  for (int row = row0; row < row1; ++row) {
    const int n = row / H, iy = row - n * H;
    // stage X row and the 3 dY rows (zero when the output row is outside the image)
    const float* xrow = X + (size_t)row * W * C9;
    for (int i = tid; i < W * 16; i += 256)
      *reinterpret_cast<float4*>(&Xs[i * 4]) = *reinterpret_cast<const float4*>(xrow + (size_t)i * 4);
    for (int a = 0; a < 3; ++a) {
      int oy = iy - (3 * khg + a) + 4;
      bool ok = (unsigned)oy < (unsigned)H;
      const float* grow = dY + ((size_t)n * H + (ok ? oy : 0)) * W * Cout;
      for (int i = tid; i < W * Cout; i += 256) Gs[a][i] = ok ? grow[i] : 0.f;
    }
    __syncthreads();
    if (wave * 32 < W) {
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const int q = wave * 32 + 2 * s + lh;
        const float a0 = Xs[q * C9 + li], a1 = Xs[q * C9 + 32 + li];
        const int ox = q - jkw + 4;
        const bool ok = jok && (unsigned)ox < (unsigned)W;
        const int gi = ok ? ox * Cout + jco : 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float b = ok ? Gs[a][gi] : 0.f;
          acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[a][0], 0, 0, 0);
          acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[a][1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // reduce the 4 waves' partial tiles through LDS, then one atomicAdd per dW element
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Red[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[a][b][r];
      __syncthreads();
      // tile element (ci_local = row, j = col): 32x32 = 1024 elements over 256 threads
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int idx = tid + 256 * e;
        int cil = idx >> 5, j = idx & 31;
        if (j < Cout * 9) {
          float v = Red[0][cil][j] + Red[1][cil][j] + Red[2][cil][j] + Red[3][cil][j];
          int co = j / 9, kw = j - co * 9, kh = 3 * khg + a;
          atomicAdd(&dW[(((size_t)co * 9 + kh) * 9 + kw) * C9 + b * 32 + cil], v);
        }
      }
      __syncthreads();
    }
}

// =======================================================================================
// bf16x3 streaming forward (precision mode 1).
//   * one wave owns a strip of OT = 24 output pixels (32 input pixels incl. the 4+4 halo) and walks DOWN the
//     image: every input row is read once (plus an 8-row halo per row range), straight from global into MFMA
//     A fragments (lane = pixel, 8 consecutive channels), split to bf16 hi/lo in registers;
//   * the 9x27x64 weights are split ONCE per block into LDS (hi/lo planes, rows j = co*9 + kw);
//   * for input row iy, tap row kh contributes to output row iy - kh + 4: a 9-slot sliding window of output-row
//     accumulators lives in registers (slot kh <-> row iy - kh + 4), shifted by one per input row, slot 8 is
//     complete and stored; all indices are compile-time;
//   * the diagonal gather  y[ox][co] += sum_kw Z[ox+kw][co*9+kw]  goes through a wave-private LDS tile: no
//     block barrier after the weight staging.
// =======================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 obf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 obf16x4;
#define WBP 72     // bf16 pitch of a weight row (64 channels + 8): conflict-free ds_read_b128 fragments
#define ZP 33
#define OT 24      // output pixels per wave strip
#define WAVES9 6

__device__ __forceinline__ void split8(const float4 a, const float4 b, obf16x8& hi, obf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  focr_split8(v, hi, lo);
}

#define C9_LOAD_ROW(IY)                                                                                   \
  {                                                                                                       \
    const bool ok_ = pxok && (unsigned)(IY) < (unsigned)H;                                                \
    const float* rp_ = xpix + (size_t)(ok_ ? (IY) : 0) * W * C9;                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                    \
      pre[2 * s_] = ok_ ? *reinterpret_cast<const float4*>(rp_ + 16 * s_) : make_float4(0.f, 0.f, 0.f, 0.f);         \
      pre[2 * s_ + 1] = ok_ ? *reinterpret_cast<const float4*>(rp_ + 16 * s_ + 4) : make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                                     \
  }

__global__ __launch_bounds__(64 * WAVES9) void conv9x9_out_fwd_bx3_kernel(
    const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias, float* __restrict__ Y,
    int H, int W, int Cout, int T, int RR, int R, int units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem9[];
  __bf16* Wh = reinterpret_cast<__bf16*>(smem9);          // [9][32][WBP]
  __bf16* Wl = Wh + 9 * 32 * WBP;
  float* Zall = reinterpret_cast<float*>(Wl + 9 * 32 * WBP);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 9 * 32 * 16; i += 64 * WAVES9) {
    const int c4 = i & 15, j = (i >> 4) & 31, kh = i >> 9;
    const int co = j / 9, kw = j - co * 9;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < Cout * 9) v = *reinterpret_cast<const float4*>(Wt + (((size_t)co * 9 + kh) * 9 + kw) * C9 + c4 * 4);
    const float a[4] = {v.x, v.y, v.z, v.w};
    obf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 hh = (__bf16)a[e];
      h[e] = hh;
      l[e] = (__bf16)(a[e] - (float)hh);
    }
    *reinterpret_cast<obf16x4*>(&Wh[(kh * 32 + j) * WBP + c4 * 4]) = h;
    *reinterpret_cast<obf16x4*>(&Wl[(kh * 32 + j) * WBP + c4 * 4]) = l;
  }
  __syncthreads();
  const int unit = blockIdx.x * WAVES9 + wave;
  if (unit >= units) return;
  const int t = unit % T, rr = (unit / T) % RR, n = unit / (T * RR);
  const int r0 = rr * R, r1 = min(H, r0 + R);
  float* Zs = Zall + wave * 32 * ZP;
  const int nout = OT * Cout;
  const int o0 = lane, o1 = lane + 64;
  const int oxl0 = o0 / Cout, co0 = o0 - oxl0 * Cout;
  const int oxl1 = o1 / Cout, co1 = o1 - oxl1 * Cout;
  const bool v0 = o0 < nout && OT * t + oxl0 < W, v1 = o1 < nout && OT * t + oxl1 < W;
  const int zb0 = o0 < nout ? oxl0 * ZP + co0 * 9 : 0, zb1 = o1 < nout ? oxl1 * ZP + co1 * 9 : 0;
  const float bs0 = bias && v0 ? bias[co0] : 0.f, bs1 = bias && v1 ? bias[co1] : 0.f;
  float w0[9], w1[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { w0[k] = 0.f; w1[k] = 0.f; }
  const int ixg = OT * t + li - 4;
  const bool pxok = (unsigned)ixg < (unsigned)W;
  const float* xpix = X + ((size_t)n * H * W + (pxok ? ixg : 0)) * C9 + 8 * lh;
  const __bf16* bhp = Wh + li * WBP + 8 * lh;
  const __bf16* blp = Wl + li * WBP + 8 * lh;
  float4 pre[8];
  C9_LOAD_ROW(r0 - 4)
  for (int iy = r0 - 4; iy < r1 + 4; ++iy) {
    const bool rowok = (unsigned)iy < (unsigned)H;
    obf16x8 ah[4], al[4];
    if (rowok) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) split8(pre[2 * s2], pre[2 * s2 + 1], ah[s2], al[s2]);
    }
    if (iy + 1 < r1 + 4) C9_LOAD_ROW(iy + 1)
    if (rowok) {
#pragma unroll
      for (int kh = 0; kh < 9; ++kh) {
        const int oy = iy - kh + 4;
        if (oy >= r0 && oy < r1) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            obf16x8 bh = *reinterpret_cast<const obf16x8*>(bhp + kh * 32 * WBP + 16 * s2);
            obf16x8 bl = *reinterpret_cast<const obf16x8*>(blp + kh * 32 * WBP + 16 * s2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s2], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s2], bh, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) Zs[((r & 3) + 8 * (r >> 2) + 4 * lh) * ZP + li] = acc[r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int kw = 0; kw < 9; ++kw) {
            s0 += Zs[zb0 + (ZP + 1) * kw];
            s1 += Zs[zb1 + (ZP + 1) * kw];
          }
          w0[kh] += s0;
          w1[kh] += s1;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    const int oyd = iy - 4;                                // slot 8 is complete
    if (oyd >= r0 && oyd < r1) {
      float* yp = Y + (((size_t)n * H + oyd) * W + OT * t) * Cout;
      if (v0) yp[o0] = w0[8] + bs0;
      if (v1) yp[o1] = w1[8] + bs1;
    }
#pragma unroll
    for (int k = 8; k > 0; --k) { w0[k] = w0[k - 1]; w1[k] = w1[k - 1]; }
    w0[0] = 0.f;
    w1[0] = 0.f;
  }
}

extern "C" int focr_colsum(const float* x, float* out, long rows, int C, int ld, hipStream_t stream);

extern "C" int focr_conv9x9_small_cout_fwd(const float* x, const float* w, const float* bias, float* y, int N,
                                           int H, int W, int Cin, int Cout, hipStream_t stream) {
  FOCR_CHECK_ARG(x && w && y, "null pointer");
  if (Cin != C9 || Cout < 1 || Cout > 3 || W + 8 > MAXT * 32 || W * Cout > 640) {
    focr_set_error("focr_conv9x9_small_cout_fwd: needs Cin == 64, Cout <= 3, W <= 152");
    return FOCR_EUNSUPPORTED;
  }
  if (focr_get_precision() != 0) {
    static const size_t lds = (size_t)2 * 9 * 32 * WBP * sizeof(__bf16) + (size_t)WAVES9 * 32 * ZP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv9x9_out_fwd_bx3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        focr_set_error("focr_conv9x9_small_cout_fwd: cannot raise the dynamic LDS limit");
        return FOCR_EHIP;
      }
      attr_set = true;
    }
    const int T = cdiv(W, OT);
    int RR = cdiv(256 * WAVES9, N * T);               // at least one strip per wave slot of the chip
    if (RR > H / 2) RR = H / 2;
    if (RR < 1) RR = 1;
    const int R = cdiv(H, RR);
    RR = cdiv(H, R);
    const int units = N * RR * T;
    hipLaunchKernelGGL(conv9x9_out_fwd_bx3_kernel, dim3(cdiv(units, WAVES9)), 64 * WAVES9, lds, stream, x, w, bias, y,
                       H, W, Cout, T, RR, R, units);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  hipLaunchKernelGGL(conv9x9_out_fwd_kernel, dim3(N * H), 320, 0, stream, x, w, bias, y, H, W, Cout);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_conv9x9_small_cout_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N,
                                             int H, int W, int Cin, int Cout, int prezeroed,
                                             hipStream_t stream) {
  FOCR_CHECK_ARG(x && dy && dw, "null pointer");
  if (Cin != C9 || Cout < 1 || Cout > 3 || W > 128 || W % 32) {
    focr_set_error("focr_conv9x9_small_cout_wgrad: needs Cin == 64, Cout <= 3, W in {32,64,96,128}");
    return FOCR_EUNSUPPORTED;
  }
  if (!prezeroed && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * 81 * C9, stream) != hipSuccess) {
    focr_set_error("focr_conv9x9_small_cout_wgrad: memset failed");
    return FOCR_EHIP;
  }
  int rows = N * H;
  int chunks = rows < 170 ? rows : 170;
  int rpc = cdiv(rows, chunks);
  chunks = cdiv(rows, rpc);
  hipLaunchKernelGGL(conv9x9_out_wgrad_kernel, dim3(3, chunks), 256, 0, stream, x, dy, dw, N, H, W, Cout, rpc);
  FOCR_LAUNCH_CHECK();
  if (dbias) return focr_colsum(dy, dbias, (long)N * H * W, Cout, Cout, stream);
  return FOCR_OK;
}

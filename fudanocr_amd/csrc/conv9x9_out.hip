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
    int H, int W, int Cout, int ldy, int T, int RR, int R, int units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem9[];
  __bf16* Wh = reinterpret_cast<__bf16*>(smem9);          // [9][32][WBP]
  __bf16* Wl = Wh + 9 * 32 * WBP;
  float* Zall = reinterpret_cast<float*>(Wl + 9 * 32 * WBP);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  {   // two-phase staging (round 5): the twelve float4 of a thread are requested together (clamped address + select
      // instead of a branch), then split and stored -- the one-pass loop cost one L2 round trip per iteration
    constexpr int IT = 9 * 32 * 16 / (64 * WAVES9);
    static_assert(9 * 32 * 16 % (64 * WAVES9) == 0, "staging trip count");
    float4 wv[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = tid + k * 64 * WAVES9;
      const int c4 = i & 15, j = (i >> 4) & 31, kh = i >> 9;
      const int jc = j < Cout * 9 ? j : 0;
      const int co = jc / 9, kw = jc - co * 9;
      wv[k] = *reinterpret_cast<const float4*>(Wt + (((size_t)co * 9 + kh) * 9 + kw) * C9 + c4 * 4);
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int i = tid + k * 64 * WAVES9;
      const int c4 = i & 15, j = (i >> 4) & 31, kh = i >> 9;
      const bool ok = j < Cout * 9;
      const float a[4] = {ok ? wv[k].x : 0.f, ok ? wv[k].y : 0.f, ok ? wv[k].z : 0.f, ok ? wv[k].w : 0.f};
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
      float* yp = Y + (((size_t)n * H + oyd) * W + OT * t) * ldy;        // ldy: pixel pitch of Y (> Cout: a channel slice)
      if (v0) yp[oxl0 * ldy + co0] = w0[8] + bs0;
      if (v1) yp[oxl1 * ldy + co1] = w1[8] + bs1;
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
  // Cout == 4 (the reference's --mask: a fourth, mask channel, main.py:31): 36 (co, kw) columns do not fit one 32-wide
  // MFMA tile and two tiles' weights (166 KB of split planes) do not fit the LDS -> two launches of two channels each,
  // writing channel slices of the 4-pitch output (split-bf16 kernel only; precision mode 0 keeps the generic kernel)
  const bool four = Cout == 4 && focr_get_precision() != 0;
  if (Cin != C9 || Cout < 1 || (Cout > 3 && !four) || W + 8 > MAXT * 32 || W * Cout > 640) {
    focr_set_error("focr_conv9x9_small_cout_fwd: needs Cin == 64, Cout <= 3 (4 in precision modes 1-3), W <= 152");
    return FOCR_EUNSUPPORTED;
  }
  if (focr_get_precision() != 0) {
    static const size_t lds = (size_t)2 * 9 * 32 * WBP * sizeof(__bf16) + (size_t)WAVES9 * 32 * ZP * sizeof(float);
    static focr_dev_flags attr_set;
    if (focr_dev_first(attr_set)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv9x9_out_fwd_bx3_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        focr_set_error("focr_conv9x9_small_cout_fwd: cannot raise the dynamic LDS limit");
        return FOCR_EHIP;
      }
      focr_dev_mark(attr_set);
    }
    const int T = cdiv(W, OT);
    int RR = cdiv(256 * WAVES9, N * T);               // at least one strip per wave slot of the chip
    if (RR > H / 2) RR = H / 2;
    if (RR < 1) RR = 1;
    const int R = cdiv(H, RR);
    RR = cdiv(H, R);
    const int units = N * RR * T;
    const int cl = four ? 2 : Cout;                   // channels per launch
    for (int c0 = 0; c0 < Cout; c0 += cl)
      hipLaunchKernelGGL(conv9x9_out_fwd_bx3_kernel, dim3(cdiv(units, WAVES9)), 64 * WAVES9, lds, stream, x,
                         w + (size_t)c0 * 81 * C9, bias ? bias + c0 : nullptr, y + c0, H, W, cl, Cout, T, RR, R, units);
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

// =======================================================================================
// bf16x3 weight gradient of the 9x9 output layer (round 4).  The fp32-MFMA kernel above re-reads every input row three
// times (one block per group of three tap rows) and runs at 1/16 of the bf16 matrix rate: 428 us on the weight-gradient
// stream at B = 128 -- the longest single launch of the step -- plus a 130 us column-sum launch for three bias values.
//   * a block walks over a range of INPUT rows; the row is split to bf16 hi / lo ONCE while it is staged, transposed
//     ([ci][pixel], two pixels per 32-bit store), so that a B fragment (8 consecutive pixels of one channel) is one
//     ds_read_b128 per plane, shared by the nine tap rows and the three waves;
//   * wave w owns tap rows kh = 3 w .. 3 w + 2: input row iy meets output row oy = iy - kh + 4; the nine dY rows (tiny:
//     W x Cout floats) are staged as fp32 with a 4-pixel zero halo, and the A fragment of lane j = co * 9 + kw is eight
//     pixels of channel co shifted by kw - 4 (stride-3 ds_read_b32, split on the fly);
//   * dW^T tile C[(co, kw)][ci] per (kh, 32-channel half): 6 accumulator tiles per wave, K = every pixel of the rows;
//   * the bias gradient falls out of the centre tap's A fragments (kw = 4 of kh = 4 sees every dY pixel exactly once);
//   * per-block partial slots + a fixed-order fold (no atomics: deterministic).
// =======================================================================================
#define W9_MAXW 128
#define W9_XTP (W9_MAXW + 8)                 // bf16 pitch of a transposed channel row: 272 B, conflict-free ds_read_b128
#define W9_GLEN ((W9_MAXW + 8) * 4)          // words of one staged dY row (4-pixel halo each side, pixel pitch <= 4); % 4 == 0
#define W9_THREADS 192

__global__ __launch_bounds__(W9_THREADS) void conv9x9_out_wgrad_bx3_kernel(const float* __restrict__ X,
                                                                           const float* __restrict__ dY,
                                                                           float* __restrict__ PART, int N, int H, int W,
                                                                           int Cout, int GST, int c0, int rows_per_block,
                                                                           long slot_floats) {
  // GST: pixel pitch of dY = its channel count; this launch takes channels c0 .. c0 + Cout - 1 of it (GST == Cout, c0 == 0
  // for the three-channel layer; the four-channel --mask layer runs as two launches of two channels: 36 (co, kw) rows do
  // not fit the 32-row M tile).  The dY rows are staged whole, at their own pitch.
  __shared__ __attribute__((aligned(16))) __bf16 Xth[C9 * W9_XTP], Xtl[C9 * W9_XTP];
  // the nine dY rows, split ONCE while staged: one 32-bit word per value = bf16 hi | bf16 lo << 16 (a lane's A fragment is eight
  // words at a 3-word stride; separating the planes costs one v_perm per word pair instead of a 24-instruction split per
  // fragment -- the fragment splits were what bounded the first version of this kernel: 1 700 VALU per wave and row)
  __shared__ __attribute__((aligned(16))) uint32_t Gs[9][W9_GLEN];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int row0 = blockIdx.x * rows_per_block, row1 = min(N * H, row0 + rows_per_block);
  const int jco = li / 9, jkw = li - jco * 9;
  const bool jok = li < Cout * 9;
  f32x16 acc[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.f;
  // A-fragment base: pixel (16 s + 8 lh + e) - kw + 4 of the halo-padded row, channel co
  // (lanes j >= Cout * 9 feed output rows that are never stored: they read what lane (co 0, kw 0) reads -- a broadcast, no
  // extra bank -- instead of a zero word of their own)
  const int abase = (jok ? (8 * lh - jkw + 8) * GST + jco : (8 * lh + 8) * GST) + c0;
  const int astep = GST, asstep = 16 * GST;
  // the halo cells (4 pixels left / right of every staged dY row) are zero for the whole launch
  for (int i = tid; i < 9 * 8 * GST; i += W9_THREADS) {
    const int kh = i / (8 * GST), g = i - kh * 8 * GST;
    Gs[kh][g < 4 * GST ? g : (W + 4) * GST + (g - 4 * GST)] = 0u;
  }
  // ---- this thread's staging items are the same for every row: up to 6 (pixel pair, 4 channels) patches of the input row
  // and up to 6 float4 of the nine dY rows (a dY row is W * GST contiguous floats)
  constexpr int DU = 6;
  const int nxi = (W / 2) * 16, qpr = W * GST / 4, ndi = 9 * qpr;
  int dkh[DU], dq[DU];
#pragma unroll
  for (int u = 0; u < DU; ++u) {
    const int i = tid + u * W9_THREADS;
    dkh[u] = i < ndi ? i / qpr : -1;
    dq[u] = i < ndi ? i - dkh[u] * qpr : 0;
  }
  float4 xa[6], xb[6], dv[DU];
  // bias gradient: channel of element 4 q of a dY row is (4 q) % GST; the per-thread sums are kept per channel SLOT
  // (slot c = channel c of dY; unused slots stay zero) and folded at the end
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  int bch[DU];
#pragma unroll
  for (int u = 0; u < DU; ++u) bch[u] = (4 * dq[u]) % GST;
  auto load_row = [&](int row) {
    const int n = row / H, iy = row - n * H;
    const float* xrow = X + (size_t)row * W * C9;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = tid + u * W9_THREADS;
      if (i < nxi) {
        // 16 adjacent lanes = 16 pixel pairs of one channel quad: their 32-bit LDS stores are consecutive words (lanes that
        // differ in the channel quad sit 4 x 68 words apart = the same bank: the first mapping -- 16 lanes over the 16
        // quads -- was an 8-way store conflict)
        const int pp = (i >> 8) * 16 + (i & 15), c4 = ((i >> 4) & 15) * 4;
        xa[u] = *reinterpret_cast<const float4*>(xrow + (size_t)(2 * pp) * C9 + c4);
        xb[u] = *reinterpret_cast<const float4*>(xrow + (size_t)(2 * pp + 1) * C9 + c4);
      }
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int oy = iy - dkh[u] + 4;
      const bool ok = dkh[u] >= 0 && (unsigned)oy < (unsigned)H;
      dv[u] = ok ? *reinterpret_cast<const float4*>(dY + ((size_t)n * H + oy) * W * GST + 4 * dq[u])
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_row = [&]() {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = tid + u * W9_THREADS;
      if (i < nxi) {
        const int pp = (i >> 8) * 16 + (i & 15), c4 = ((i >> 4) & 15) * 4;
        const float a0[4] = {xa[u].x, xa[u].y, xa[u].z, xa[u].w}, a1[4] = {xb[u].x, xb[u].y, xb[u].z, xb[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          focr_bf16x2 hh, ll;
          focr_split2(f32x2{a0[e], a1[e]}, hh, ll);
          *reinterpret_cast<focr_bf16x2*>(&Xth[(c4 + e) * W9_XTP + 2 * pp]) = hh;
          *reinterpret_cast<focr_bf16x2*>(&Xtl[(c4 + e) * W9_XTP + 2 * pp]) = ll;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < DU; ++u)
      if (dkh[u] >= 0) {
        focr_bf16x2 h0, l0, h1, l1;
        focr_split2(f32x2{dv[u].x, dv[u].y}, h0, l0);
        focr_split2(f32x2{dv[u].z, dv[u].w}, h1, l1);
        const uint32_t ha = __builtin_bit_cast(uint32_t, h0), la = __builtin_bit_cast(uint32_t, l0);
        const uint32_t hb = __builtin_bit_cast(uint32_t, h1), lb = __builtin_bit_cast(uint32_t, l1);
        *reinterpret_cast<uint4*>(&Gs[dkh[u]][4 * GST + 4 * dq[u]]) =
            make_uint4(__builtin_amdgcn_perm(la, ha, 0x05040100u), __builtin_amdgcn_perm(la, ha, 0x07060302u),
                       __builtin_amdgcn_perm(lb, hb, 0x05040100u), __builtin_amdgcn_perm(lb, hb, 0x07060302u));
        if (dkh[u] == 4) {                                // tap row kh = 4 is dY row iy itself: every pixel exactly once
          const float ev[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
          int c = bch[u];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bs[0] += c == 0 ? ev[e] : 0.f;
            bs[1] += c == 1 ? ev[e] : 0.f;
            bs[2] += c == 2 ? ev[e] : 0.f;
            bs[3] += c == 3 ? ev[e] : 0.f;
            c = c + 1 == GST ? 0 : c + 1;
          }
        }
      }
  };

  if (row0 < row1) load_row(row0);
  for (int row = row0; row < row1; ++row) {
    store_row();                                        // this row (requested during the previous row's products)
    __syncthreads();
    if (row + 1 < row1) load_row(row + 1);
    for (int s = 0; s < W / 16; ++s) {
      obf16x8 bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bh[t] = *reinterpret_cast<const obf16x8*>(&Xth[(li + 32 * t) * W9_XTP + 16 * s + 8 * lh]);
        bl[t] = *reinterpret_cast<const obf16x8*>(&Xtl[(li + 32 * t) * W9_XTP + 16 * s + 8 * lh]);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const uint32_t* g = &Gs[3 * wave + a][abase + s * asstep];     // 8 words at a Cout-word stride
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = g[e * astep];
        typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
        const u32x4 hw = {__builtin_amdgcn_perm(w[1], w[0], 0x05040100u), __builtin_amdgcn_perm(w[3], w[2], 0x05040100u),
                          __builtin_amdgcn_perm(w[5], w[4], 0x05040100u), __builtin_amdgcn_perm(w[7], w[6], 0x05040100u)};
        const u32x4 lw = {__builtin_amdgcn_perm(w[1], w[0], 0x07060302u), __builtin_amdgcn_perm(w[3], w[2], 0x07060302u),
                          __builtin_amdgcn_perm(w[5], w[4], 0x07060302u), __builtin_amdgcn_perm(w[7], w[6], 0x07060302u)};
        const obf16x8 ah = __builtin_bit_cast(obf16x8, hw), al = __builtin_bit_cast(obf16x8, lw);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[a][t], 0, 0, 0);
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[a][t], 0, 0, 0);
          acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[a][t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // ---- this block's partial: slot [Cout][9][9][64] (+ Cout bias sums), plain stores
  float* slot = PART + (size_t)blockIdx.x * slot_floats;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int kh = 3 * wave + a;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (j < Cout * 9) {
          const int co = (j * 57) >> 9, kw = j - 9 * co;       // j / 9 for j < 32
          slot[(((size_t)co * 9 + kh) * 9 + kw) * C9 + li + 32 * t] = acc[a][t][r];
        }
      }
  }
  // bias sums: per-thread channel sums -> block (Xth is free now)
  {
    float* red = reinterpret_cast<float*>(Xth);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) red[c * W9_THREADS + tid] = bs[c];
    __syncthreads();
    if (tid < Cout) {
      float t = 0.f;
      for (int i = 0; i < W9_THREADS; ++i) t += red[(c0 + tid) * W9_THREADS + i];
      slot[(size_t)Cout * 81 * C9 + tid] = t;
    }
  }
}

// dst = sum over the slots in slot order (fixed order: deterministic); block = 8 float4 elements x 32 slot groups
__global__ __launch_bounds__(256) void conv9x9_out_wgrad_fold_kernel(const float* __restrict__ PART, float* __restrict__ dw,
                                                                     float* __restrict__ dbias, long n_dw4,
                                                                     long slot_floats, int nslots, int Cout) {
  __shared__ float4 red[32][8];
  const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + e;
  float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int per = (nslots + 31) / 32, s0 = g * per, s1 = min(nslots, s0 + per);
  if (i < n_dw4) {
    for (int b = s0; b < s1; ++b) {
      const float4 v = reinterpret_cast<const float4*>(PART + (size_t)b * slot_floats)[i];
      sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
    }
  } else if (i == n_dw4 && dbias) {                      // the bias sums: element i == n_dw4 of every slot (Cout floats)
    for (int b = s0; b < s1; ++b) {
      const float* v = PART + (size_t)b * slot_floats + n_dw4 * 4;
      sacc.x += v[0];
      if (Cout > 1) sacc.y += v[1];
      if (Cout > 2) sacc.z += v[2];
      if (Cout > 3) sacc.w += v[3];
    }
  }
  red[g][e] = sacc;
  __syncthreads();
  if (threadIdx.x < 8) {
    float4 t = red[0][e];
#pragma unroll
    for (int q = 1; q < 32; ++q) { t.x += red[q][e].x; t.y += red[q][e].y; t.z += red[q][e].z; t.w += red[q][e].w; }
    if (i < n_dw4) {
      reinterpret_cast<float4*>(dw)[i] = t;
    } else if (i == n_dw4 && dbias) {
      dbias[0] = t.x;
      if (Cout > 1) dbias[1] = t.y;
      if (Cout > 2) dbias[2] = t.z;
      if (Cout > 3) dbias[3] = t.w;
    }
  }
}

static void w9_plan(int N, int H, int& nb, int& rpb) {
  const int rows = N * H;
  nb = rows < 512 ? rows : 512;           // two 3-wave blocks per CU: one block alone leaves a SIMD idle and nothing to cover its LDS gathers
  rpb = cdiv(rows, nb);
  nb = cdiv(rows, rpb);
}
extern "C" long focr_conv9x9_small_cout_wgrad_ws_floats(int N, int H, int W, int Cout) {
  int nb, rpb;
  w9_plan(N, H, nb, rpb);
  return (long)nb * ((long)Cout * 81 * C9 + 4);
}
// as focr_conv9x9_small_cout_wgrad, on the bf16 matrix pipe with split operands (precision modes 1-3) and with the bias
// gradient from the same pass.  dw / dbias are OVERWRITTEN.  ws: focr_conv9x9_small_cout_wgrad_ws_floats() floats.
extern "C" int focr_conv9x9_small_cout_wgrad_ws(const float* x, const float* dy, float* dw, float* dbias, float* ws,
                                                long ws_floats, int N, int H, int W, int Cin, int Cout,
                                                hipStream_t stream) {
  FOCR_CHECK_ARG(x && dy && dw && ws, "null pointer");
  if (Cin != C9 || Cout < 1 || Cout > 4 || W > W9_MAXW || W % 32 || focr_get_precision() == 0) {
    focr_set_error("focr_conv9x9_small_cout_wgrad_ws: needs Cin == 64, Cout <= 4, W %% 32 == 0, W <= 128, precision != 0");
    return FOCR_EUNSUPPORTED;
  }
  int nb, rpb;
  w9_plan(N, H, nb, rpb);
  FOCR_CHECK_ARG(ws_floats >= (long)nb * ((long)Cout * 81 * C9 + 4), "workspace too small");
  // Cout == 4 (--mask): two launches of two channels each (see the kernel), the workspace re-used in stream order
  const int cl = Cout == 4 ? 2 : Cout;
  const long slot = (long)cl * 81 * C9 + 4;
  const long n_dw4 = (long)cl * 81 * C9 / 4;
  for (int c0 = 0; c0 < Cout; c0 += cl) {
    hipLaunchKernelGGL(conv9x9_out_wgrad_bx3_kernel, dim3(nb), W9_THREADS, 0, stream, x, dy, ws, N, H, W, cl, Cout, c0, rpb,
                       slot);
    hipLaunchKernelGGL(conv9x9_out_wgrad_fold_kernel, dim3((int)((n_dw4 + 1 + 7) / 8)), 256, 0, stream, (const float*)ws,
                       dw + (size_t)c0 * 81 * C9, dbias ? dbias + c0 : nullptr, n_dw4, slot, nb, cl);
  }
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

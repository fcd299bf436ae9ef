// Implicit-GEMM NHWC convolution / linear on the bf16 MFMA pipe with split operands ("bf16x3"):
// same tiling, gather and epilogue as conv_igemm.hip's VEC path (Cin % 32 == 0), but the LDS tiles
// hold bf16 hi/lo pairs (x = hi + lo) and each 16-deep k-step issues
//   A_hi.B_hi + A_hi.B_lo + A_lo.B_hi   on v_mfma_f32_32x32x16_bf16  (fp32 accumulate).
// End-to-end error equals plain fp32's (tools/exp_split_precision.py); MFMA time is 3/16 of the
// f32-MFMA kernel's, so these layers become L2/HBM-traffic bound instead.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define XBM 128
#define XBK 32
#define XRP 40   // LDS row pitch in bf16 (32 + 8): 80 B, conflict-free ds_read_b128 fragments

struct ConvGeomX {
  int N, H, W, Cin, OH, OW, Cout, KH, KW, padH, padW, Ktot, M, ldy, ldr, ldx;
};

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) {
  const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    __bf16 h = (__bf16)a[e];
    hi[e] = h;
    lo[e] = (__bf16)(a[e] - (float)h);
  }
}

template <int NT>
__global__ __launch_bounds__(256) void conv_fwd_bx3_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ R, float* __restrict__ Y,
                                                           ConvGeomX g, float alpha, int relu) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) __bf16 Ah[XBM * XRP], Al[XBM * XRP];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[BN * XRP], Bl[BN * XRP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * XBM, n0 = blockIdx.y * BN;
  const int nchunks = g.Ktot / XBK;

  // staging: thread owns float4 column c4 = tid&7 of rows (tid>>3) + 32*i
  int rbase[4], riy[4], rix[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int p = m0 + (tid >> 3) + 32 * i;
    if (p < g.M) {
      int n = p / (g.OH * g.OW);
      int rem = p - n * (g.OH * g.OW);
      int oy = rem / g.OW, ox = rem - oy * g.OW;
      riy[i] = oy - g.padH;
      rix[i] = ox - g.padW;
      rbase[i] = n * g.H * g.W;
    } else {
      riy[i] = -100000;
      rix[i] = 0;
      rbase[i] = 0;
    }
  }
  constexpr int BV = (BN * XBK / 4) / 256;
  float4 areg[4], breg[BV];
  auto load_chunk = [&](int c) {
    const int k0 = c * XBK;
    int tap = k0 / g.Cin;
    int ci0 = k0 - tap * g.Cin + (tid & 7) * 4;
    int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = riy[i] + kh, ix = rix[i] + kw;
      bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(X + ((size_t)(rbase[i] + iy * g.W + ix) * g.ldx + ci0));
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      int idx = tid + 256 * i;
      int co = n0 + (idx >> 3);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (co < g.Cout) v = *reinterpret_cast<const float4*>(Wt + (size_t)co * g.Ktot + k0 + (idx & 7) * 4);
      breg[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x4 h, l;
      split4(areg[i], h, l);
      int o = ((tid >> 3) + 32 * i) * XRP + (tid & 7) * 4;
      *reinterpret_cast<bf16x4*>(&Ah[o]) = h;
      *reinterpret_cast<bf16x4*>(&Al[o]) = l;
    }
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      bf16x4 h, l;
      split4(breg[i], h, l);
      int idx = tid + 256 * i;
      int o = (idx >> 3) * XRP + (idx & 7) * 4;
      *reinterpret_cast<bf16x4*>(&Bh[o]) = h;
      *reinterpret_cast<bf16x4*>(&Bl[o]) = l;
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int aoff = (wave * 32 + li) * XRP + 8 * lh;
  const int boff = li * XRP + 8 * lh;

  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Ah[aoff + 16 * m]);
      bf16x8 al = *reinterpret_cast<const bf16x8*>(&Al[aoff + 16 * m]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Bh[boff + nt * 32 * XRP + 16 * m]);
        bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Bl[boff + nt * 32 * XRP + 16 * m]);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int co = n0 + nt * 32 + li;
    if (co >= g.Cout) continue;
    float b = bias ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int p = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (p < g.M) {
        float v = alpha * acc[nt][r] + b;
        if (R) v += R[(size_t)p * g.ldr + co];
        if (relu) v = fmaxf(v, 0.f);
        Y[(size_t)p * g.ldy + co] = v;
      }
    }
  }
}

// launcher used by focr_conv2d_fwd (conv_igemm.hip) when precision == bf16x3 and the layer is VEC-capable
int focr_conv_fwd_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y,
                      int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int padH, int padW,
                      int M, int ldy, int ldr, int ldx, float alpha, int relu, hipStream_t stream) {
  ConvGeomX g{N, H, W, Cin, OH, OW, Cout, KH, KW, padH, padW, KH * KW * Cin, M, ldy, ldr, ldx};
  bool wide = Cout > 32;
  dim3 grid((M + XBM - 1) / XBM, (Cout + (wide ? 63 : 31)) / (wide ? 64 : 32));
  if (wide)
    hipLaunchKernelGGL((conv_fwd_bx3_kernel<2>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  else
    hipLaunchKernelGGL((conv_fwd_bx3_kernel<1>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  return 0;
}

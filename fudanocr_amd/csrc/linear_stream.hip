// Streaming 1x1 conv / Linear for short contractions (K = 64 or 128), bf16x3: the transformer linears of the TBSRN
// FeatureEnhancer (tbsrn.py:74,91,109-130,162-163: Q/K/V/O projections, FFN, 128->64) and their data gradients.
//
// These layers are HBM-bound (read [M,K], write [M,N], M = B*1024 rows, a 64 KB weight matrix), but the tiled
// implicit-GEMM kernel runs them at ~1.6x the streaming time: every block walks load -> barrier -> MFMA -> barrier
// -> store in lockstep and re-stages the weights per 32-deep chunk.  Here
//   * the block splits its [32*NT][K] weight slice to bf16 hi/lo ONCE and keeps it in LDS;
//   * each wave independently streams 32-row tiles: the A fragments come straight from global memory in MFMA layout
//     (the loads of the next tile are in flight while the current one is multiplied: K/16 * NT * 3 MFMAs), so the
//     main loop has no barrier and no LDS traffic besides the weight fragments; two blocks (8 waves) per CU
//     (a first version staged A through wave-private LDS tiles with one wave per SIMD: the exposed ds_read latency
//     made it no faster than the tiled kernel);
//   * epilogue (alpha, bias, residual, relu) as in conv_fwd_bx3_kernel.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 lbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 lbf16x4;

__device__ __forceinline__ void lsplit4(float4 v, lbf16x4& hi, lbf16x4& lo) { focr_split4(v, hi, lo); }

__device__ __forceinline__ void lsplit8(const float4 a, const float4 b, lbf16x8& hi, lbf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  focr_split8(v, hi, lo);
}

template <int K, int NT, bool HAS_RES, bool HAS_DROP>
__global__ __launch_bounds__(256, 2) void linear_stream_bx3_kernel(const float* __restrict__ X,
                                                                   const float* __restrict__ Wt,   // [Cout][K]
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ R, float* __restrict__ Y,
                                                                   int M, int Cout, int ldx, int ldy, int ldr,
                                                                   float alpha, int relu, uint32_t drop_k,
                                                                   float drop_scale, uint32_t drop_seed,
                                                                   float mask_scale,
                                                                   const uint64_t* __restrict__ epoch) {
  if (HAS_DROP) drop_seed = focr_epoch_seed32(drop_seed, epoch);
  constexpr int KP = K + 8;            // bf16 pitch: conflict-free ds_read_b128 fragment reads
  constexpr int Q = K / 4;             // float4 per weight row
  constexpr int KS = K / 16;           // MFMA k-steps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ls[];
  __bf16* Wh = reinterpret_cast<__bf16*>(smem_ls);         // [32*NT][KP]
  __bf16* Wl = Wh + 32 * NT * KP;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32 * NT;
  for (int i = tid; i < 32 * NT * Q; i += 256) {
    const int row = i / Q, c4 = i - row * Q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n0 + row < Cout) v = *reinterpret_cast<const float4*>(Wt + (size_t)(n0 + row) * K + 4 * c4);
    lbf16x4 h, l;
    lsplit4(v, h, l);
    *reinterpret_cast<lbf16x4*>(&Wh[row * KP + 4 * c4]) = h;
    *reinterpret_cast<lbf16x4*>(&Wl[row * KP + 4 * c4]) = l;
  }
  __syncthreads();
  const int ntiles = (M + 31) / 32;
  const int stride = gridDim.x * 4;
  // A fragments straight from global in MFMA layout: lane (li, lh) owns row li, k = 16 s + 8 lh .. + 7 (32 bytes per
  // k-step); the 2*KS loads of the NEXT tile are issued as one batch right after the current tile was split, so they
  // cover each 512-byte row completely while its lines are still in the vector L1
  float4 rg[2 * KS];
#define LS_LOAD(T)                                                                              \
  {                                                                                             \
    const int p_ = (T) * 32 + li;                                                               \
    const float* xp_ = X + (size_t)(p_ < M ? p_ : 0) * ldx + 8 * lh;                            \
    _Pragma("unroll") for (int s_ = 0; s_ < KS; ++s_) {                                         \
      rg[2 * s_] = *reinterpret_cast<const float4*>(xp_ + 16 * s_);                             \
      rg[2 * s_ + 1] = *reinterpret_cast<const float4*>(xp_ + 16 * s_ + 4);                     \
    }                                                                                           \
  }
  int t = blockIdx.x * 4 + wave;
  if (t < ntiles) LS_LOAD(t)
  float bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int co = n0 + nt * 32 + li;
    bv[nt] = (bias && co < Cout) ? bias[co] : 0.f;
  }
  for (; t < ntiles; t += stride) {
    lbf16x8 ah[KS], al[KS];
    const bool rowok = t * 32 + li < M;          // rows past the end were loaded from row 0: zero them
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      lsplit8(rowok ? rg[2 * s] : z4, rowok ? rg[2 * s + 1] : z4, ah[s], al[s]);
    }
    const int tn = t + stride;
    if (tn < ntiles) LS_LOAD(tn)
    // The weight fragments are the same for every tile: left alone, the compiler hoists all 2*NT*KS ds_reads out of
    // the tile loop (256 VGPRs, spills).  An opaque per-iteration zero in the address keeps them inside.
    int zofs;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zofs));
    const __bf16* whp = Wh + zofs;
    const __bf16* wlp = Wl + zofs;
    // two column tiles at a time (32 accumulator registers live): the A fragments are reused for every pair
#pragma unroll
    for (int ng = 0; ng < NT; ng += 2) {
      f32x16 acc[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
      // residual values of this column pair: in flight during the MFMAs instead of exposed in the epilogue
      float rv[2][16];
      if (HAS_RES) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int co = n0 + (ng + u) * 32 + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int p = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            rv[u][r] = (p < M && co < Cout) ? R[(size_t)p * ldr + co] : 0.f;
          }
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          lbf16x8 bh = *reinterpret_cast<const lbf16x8*>(&whp[((ng + u) * 32 + li) * KP + 16 * s + 8 * lh]);
          lbf16x8 bl = *reinterpret_cast<const lbf16x8*>(&wlp[((ng + u) * 32 + li) * KP + 16 * s + 8 * lh]);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc[u], 0, 0, 0);
          acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc[u], 0, 0, 0);
        }
      }
      // fused dropout (FFN: Dropout(relu(w_1 x)), tbsrn.py:162-163): 32 Bernoulli keep bits per lane and column pair,
      // bit-sliced from a xorshift stream (P(keep) = drop_k / 65536).  The backward never needs these bits: a dropped
      // element IS a zero of y, so relu_bwd_scaled on (y > 0) with scale 1/P(keep) is exact.
      uint32_t keep = 0xffffffffu;
      if (HAS_DROP) {
        uint32_t x = hash32(drop_seed ^ hash32((uint32_t)(t * (NT / 2) + (ng >> 1)) * 0x9E3779B1U + blockIdx.y) ^
                            (uint32_t)lane * 0x85EBCA6BU) | 1u;
        keep = 0u;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          x ^= x << 13; x ^= x >> 17; x ^= x << 5;
          keep = ((drop_k >> b) & 1u) ? (keep | x) : (keep & x);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int co = n0 + (ng + u) * 32 + li;
        if (co >= Cout) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int p = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (p < M) {
            float v = alpha * acc[u][r] + bv[ng + u];
            // mask_scale != 0: R is not added but GATES the result (relu / relu-dropout backward fused into the data
            // gradient of the FOLLOWING linear: g = (h > 0) ? scale * (dy W) : 0, R = h)
            if (HAS_RES) v = mask_scale != 0.f ? (rv[u][r] > 0.f ? v * mask_scale : 0.f) : v + rv[u][r];
            if (relu) v = fmaxf(v, 0.f);
            if (HAS_DROP) v = ((keep >> (u * 16 + r)) & 1u) ? v * drop_scale : 0.f;
            Y[(size_t)p * ldy + co] = v;
          }
        }
      }
    }
  }
}

template <int K, int NT, bool HAS_RES, bool HAS_DROP>
static int launch_ls_(const float* x, const float* w, const float* bias, const float* r, float* y, int M, int Cout,
                      int ldx, int ldy, int ldr, float alpha, int relu, uint32_t drop_k, float drop_scale,
                      uint32_t drop_seed, float mask_scale, hipStream_t stream) {
  const size_t lds = (size_t)2 * 32 * NT * (K + 8) * sizeof(__bf16);
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_stream_bx3_kernel<K, NT, HAS_RES, HAS_DROP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 0;
    focr_dev_mark(attr_set);
  }
  const int ny = cdiv(Cout, 32 * NT);
  const int ntiles = cdiv(M, 32);
  int nb = 2 * 256;                                    // two blocks (8 waves) per CU: launch_bounds(256, 2)
  if (nb > cdiv(ntiles, 4)) nb = cdiv(ntiles, 4);
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((linear_stream_bx3_kernel<K, NT, HAS_RES, HAS_DROP>), dim3(nb, ny), 256, lds, stream, x, w, bias,
                     r, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale, drop_seed, mask_scale, focr_seed_epoch());
  return 1;
}

template <int K, int NT>
static int launch_ls(const float* x, const float* w, const float* bias, const float* r, float* y, int M, int Cout,
                     int ldx, int ldy, int ldr, float alpha, int relu, uint32_t drop_k, float drop_scale,
                     uint32_t drop_seed, float mask_scale, hipStream_t stream) {
  if (drop_k) {
    if (r) return 0;                                   // dropout + residual: not built
    return launch_ls_<K, NT, false, true>(x, w, bias, r, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale,
                                          drop_seed, 0.f, stream);
  }
  if (mask_scale != 0.f && !r) return 0;
  return r ? launch_ls_<K, NT, true, false>(x, w, bias, r, y, M, Cout, ldx, ldy, ldr, alpha, relu, 0u, 1.f, 0u, mask_scale,
                                            stream)
           : launch_ls_<K, NT, false, false>(x, w, bias, r, y, M, Cout, ldx, ldy, ldr, alpha, relu, 0u, 1.f, 0u, 0.f,
                                             stream);
}

#ifndef LS_MIN_ROWS
#define LS_MIN_ROWS 16384
#endif

// used by focr_conv2d_fwd (conv_igemm.hip) for 1x1 layers; returns 1 if the layer was handled here
int focr_linear_stream_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y, int M,
                           int Cin, int Cout, int ldx, int ldy, int ldr, float alpha, int relu, uint32_t drop_k,
                           float drop_scale, uint32_t drop_seed, float mask_scale, hipStream_t stream) {
  if (M < LS_MIN_ROWS || ldx % 4 || Cout % 32) return 0;
  if (Cin == 128) {
    if (Cout % 128 == 0) return launch_ls<128, 4>(x, w, bias, residual, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale,
                                                 drop_seed, mask_scale, stream);
    if (Cout % 64 == 0) return launch_ls<128, 2>(x, w, bias, residual, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale,
                                                 drop_seed, mask_scale, stream);
  } else if (Cin == 64) {
    if (Cout % 128 == 0) return launch_ls<64, 4>(x, w, bias, residual, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale,
                                                 drop_seed, mask_scale, stream);
    if (Cout % 64 == 0) return launch_ls<64, 2>(x, w, bias, residual, y, M, Cout, ldx, ldy, ldr, alpha, relu, drop_k, drop_scale,
                                                 drop_seed, mask_scale, stream);
  }
  return 0;
}

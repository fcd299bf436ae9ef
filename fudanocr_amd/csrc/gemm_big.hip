// Large plain GEMMs of the recognizers (1 x 1 "convolutions" = nn.Linear on many rows):  Y[M][N] = alpha X[M][K] W[N][K]^T
// (+ bias, + residual, relu), split-bf16 products, fp32 accumulate.
//
// Where: the cross-attention K / V projections of the focus-loss recognizers (loss/transformer.py:232-246: 1024 -> 1024 on
// B x 256 memory positions = 32 768 rows at B = 128, forward on HR and SR + the SR data gradient: six launches per step) and
// the stroke-level-decomposition decoder's (model/transformer.py of that project, 8 192 rows).  On the generic implicit-GEMM
// kernel (conv_bx3.hip: 128 x 64 tile, 32-deep chunks, two barriers per chunk) a block moves 24 KB through L2 per 524 kflop
// = 21.8 flop/B: 4 096 blocks x 768 KB = 3.1 GB for 68.7 GFLOP -- 295 us, bound by the L2, not by the matrix pipe.
//   * 256 x 128 tile, eight waves as 4 x 2, a wave owns 64 x 64 = four 32 x 32 accumulators: 43.7 flop per L2 byte;
//   * the N tiles of one row block are neighbouring blocks (blockIdx.x = n tile): its 256 x K slab of X comes out of HBM
//     once and is an L2 hit for the other N / 128 - 1 blocks; the weights (N x K, a few MB) stay L2 / Infinity-Cache resident;
//   * operand tiles split to bf16 hi / lo while they are stored to LDS (pitch 40: conflict-free ds_read_b128 fragments),
//     two LDS stages, the next chunk's global loads in flight in registers during the current chunk's 24 MFMAs per wave,
//     ONE barrier per chunk.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 gb_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 gb_bf16x4;

#define GB_BM 256
#define GB_BN 128
#define GB_BK 32
#define GB_RP 40                                    // LDS row pitch in bf16 (32 + 8)
#define GB_A_PLANE (GB_BM * GB_RP * 2)              // bytes of one plane of the A tile: 20 480
#define GB_B_PLANE (GB_BN * GB_RP * 2)              // 10 240
#define GB_STAGE (2 * GB_A_PLANE + 2 * GB_B_PLANE)  // 61 440
#define GB_LDS (2 * GB_STAGE)                       // 122 880: one block of eight waves per CU

__global__ __launch_bounds__(512, 1) void gemm_big_bx3_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                              const float* __restrict__ bias, const float* __restrict__ R,
                                                              float* __restrict__ Y, int M, int K, int N, int ldx, int ldy,
                                                              int ldr, float alpha, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * GB_BN, m0 = blockIdx.y * GB_BM;
  const int nchunks = K / GB_BK;

  // staging: thread owns float4 column c4 = tid & 7 of rows (tid >> 3) + 64 i: A i = 0..3, B i = 0..1
  const int c4 = tid & 7, r0 = tid >> 3;
  const float* ap[4];
  const float* bp[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = m0 + r0 + 64 * i;
    if (row >= M) row = M - 1;                        // clamped rows are computed and never stored
    ap[i] = X + (size_t)row * ldx + c4 * 4;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) bp[i] = Wt + (size_t)(n0 + r0 + 64 * i) * K + c4 * 4;
  float4 areg[4], breg[2];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = *reinterpret_cast<const float4*>(ap[i] + c * GB_BK);
#pragma unroll
    for (int i = 0; i < 2; ++i) breg[i] = *reinterpret_cast<const float4*>(bp[i] + c * GB_BK);
  };
  auto store_chunk = [&](int buf) {
    unsigned char* st = gb_smem + buf * GB_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      gb_bf16x4 h, l;
      focr_split4(areg[i], h, l);
      const int o = ((r0 + 64 * i) * GB_RP + c4 * 4) * 2;
      *reinterpret_cast<gb_bf16x4*>(st + o) = h;
      *reinterpret_cast<gb_bf16x4*>(st + GB_A_PLANE + o) = l;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      gb_bf16x4 h, l;
      focr_split4(breg[i], h, l);
      const int o = ((r0 + 64 * i) * GB_RP + c4 * 4) * 2;
      *reinterpret_cast<gb_bf16x4*>(st + 2 * GB_A_PLANE + o) = h;
      *reinterpret_cast<gb_bf16x4*>(st + 2 * GB_A_PLANE + GB_B_PLANE + o) = l;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int aoff = ((wm * 64 + li) * GB_RP + 8 * lh) * 2;
  const int boff = 2 * GB_A_PLANE + ((wn * 64 + li) * GB_RP + 8 * lh) * 2;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    if (more) load_chunk(c + 1);
    const unsigned char* st = gb_smem + (c & 1) * GB_STAGE;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      gb_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const gb_bf16x8*>(st + aoff + t * 32 * GB_RP * 2 + 32 * m);
        al[t] = *reinterpret_cast<const gb_bf16x8*>(st + GB_A_PLANE + aoff + t * 32 * GB_RP * 2 + 32 * m);
        bh[t] = *reinterpret_cast<const gb_bf16x8*>(st + boff + t * 32 * GB_RP * 2 + 32 * m);
        bl[t] = *reinterpret_cast<const gb_bf16x8*>(st + GB_B_PLANE + boff + t * 32 * GB_RP * 2 + 32 * m);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
    // (the other stage: last read in iteration c - 1, every wave is past that barrier.  Storing between the two k-steps, under
    // the first one's MFMAs, measured 3-8 % SLOWER: profiles/r06_gemm_big_store_mid.txt)
    if (more) store_chunk((c + 1) & 1);
    __syncthreads();
  }
  // C tile (a, b) of this wave: rows m0 + 64 wm + 32 a + (r & 3) + 8 (r >> 2) + 4 lh, column n0 + 64 wn + 32 b + li
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int co = n0 + wn * 64 + 32 * b + li;
    const float bs = bias ? bias[co] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = m0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < M) {
          float v = alpha * acc[a][b][r] + bs;
          if (R) v += R[(size_t)p * ldr + co];
          if (relu) v = fmaxf(v, 0.f);
          Y[(size_t)p * ldy + co] = v;
        }
      }
  }
}

// used by conv2d_fwd_impl (conv_igemm.hip) for 1 x 1 layers; returns 1 if the layer was handled here
int focr_gemm_big_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y, long M, int K,
                      int N, int ldx, int ldy, int ldr, float alpha, int relu, hipStream_t stream) {
  static const bool on = !(getenv("FOCR_GEMM_BIG") && getenv("FOCR_GEMM_BIG")[0] == '0');
  // enough 256 x 128 tiles to fill the chip, a contraction worth the pipeline, aligned 16-byte rows
  if (!on || K % GB_BK || K < 256 || N % GB_BN || M > (1l << 30) || (M + GB_BM - 1) / GB_BM * (N / GB_BN) < 192 ||
      ldx % 4 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15))
    return 0;
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_bx3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            GB_LDS) != hipSuccess)
      return 0;
    focr_dev_mark(attr_set);
  }
  dim3 grid(N / GB_BN, (int)((M + GB_BM - 1) / GB_BM));
  hipLaunchKernelGGL(gemm_big_bx3_kernel, grid, 512, GB_LDS, stream, x, w, bias, residual, y, (int)M, K, N, ldx, ldy, ldr, alpha,
                     relu);
  return 1;
}

// Weight (+ bias) gradient of the transformer linears:  dW[co][k] = sum_p dY[p][co] X[p][k],  db[co] = sum_p dY[p][co]
// with p over the B*H*W token rows (131072 at the bench shape) and co, k = 128 (384 for the packed QKV projection).
//
// The contraction runs over the SLOW memory axis of both operands, so this is a pure streaming problem: 134 MB read
// for 64 KB of result (algorithmic 1024 B per token row), ~17 us at the HBM roofline, 13 GFLOP x 3 products of MFMA
// work (5 us).  The generic kernel (conv_bx3.hip conv_wgrad_bx3_wide_kernel) moves every element through VGPRs, splits
// it, writes a transposed bf16 copy to LDS behind two barriers per 64 rows with one block per CU: 109 us.  Here
//   * rows are DMA'd straight into LDS as fp32 (global_load_lds, 16 B per lane, no VGPRs), four 16-row stages per
//     block, two blocks per CU: ~96 KB per CU in flight, ONE barrier per stage;
//   * no transpose pass: the reduction index of the MFMA is the row, so a lane's eight k-values of a fragment are eight
//     rows of ONE column -- read as ds_read_b64 (two neighbouring columns = the same lane of two interleaved 32-wide
//     tiles: tile t of a wave holds columns 2 i + t), split to bf16 hi/lo in registers, three MFMAs per tile pair;
//   * the bias gradient falls out of the A fragments (fp32 adds of the raw values);
//   * every row split writes its 128 x 128 tile to its own slot with plain stores; slot_reduce_kernel folds the slots
//     in a fixed order (deterministic, no atomics) straight into the gradient buffer.
#include <stdlib.h>
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 lw_bf16x8;
typedef __attribute__((ext_vector_type(16))) float lw_f32x16;
typedef __attribute__((ext_vector_type(2))) float lw_f32x2;

#define LW_ROWS 16                      // rows per stage = one MFMA k-step
#define LW_PIECE 1056                   // bytes of one DMA piece (2 rows x 128 fp32) + 32 B: rows r and r + 8 (the two
                                        // lane halves of a fragment read) land 32 banks apart
#define LW_MAT (8 * LW_PIECE)           // one operand's 16 rows
#define LW_STAGE (2 * LW_MAT)           // dY rows, then X rows
#define LW_NS 4
#define LW_LDS (LW_NS * LW_STAGE)       // 67584 B: two blocks per CU

struct LwRaw {
  lw_f32x2 a[8], b[8];                  // [row j of this lane half]: columns (2 li, 2 li + 1) of dY / X
};
template <int OFF>
__device__ __forceinline__ void lw_ldsr(lw_f32x2& d, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
__device__ __forceinline__ void lw_read(LwRaw& r, unsigned aaddr, unsigned baddr) {
#define LW_RD(J)                                                      \
  lw_ldsr<((J) >> 1) * LW_PIECE + ((J) & 1) * 512>(r.a[J], aaddr);    \
  lw_ldsr<((J) >> 1) * LW_PIECE + ((J) & 1) * 512>(r.b[J], baddr);
  LW_RD(0) LW_RD(1) LW_RD(2) LW_RD(3) LW_RD(4) LW_RD(5) LW_RD(6) LW_RD(7)
#undef LW_RD
}
__device__ __forceinline__ void lw_wait(LwRaw& r) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.a[4]), "+v"(r.a[5]), "+v"(r.a[6]),
                 "+v"(r.a[7]), "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]), "+v"(r.b[4]), "+v"(r.b[5]),
                 "+v"(r.b[6]), "+v"(r.b[7]));
}
// eight fp32 rows of one column -> bf16 hi / lo fragments
__device__ __forceinline__ void lw_split(const lw_f32x2 (&v)[8], int t, lw_bf16x8& hi, lw_bf16x8& lo) {
  const float x[8] = {v[0][t], v[1][t], v[2][t], v[3][t], v[4][t], v[5][t], v[6][t], v[7][t]};
  focr_split8(x, hi, lo);
}

// grid (K / 128, Cout / 128, splits), 256 threads.  PART: [splits][Cout * K + Cout] floats.
// HALF (round 5): Cout = 64 (the FeatureEnhancer's 128 -> 64 projection, which ran on the generic kernel at 47 us): a dY row
// is 256 bytes, so the two lane quarters of a DMA piece fetch the same 64 columns (the LDS image keeps the 512-byte row
// slots, columns 64.. are never read), all four waves read dY columns 0..63, and the wave pair (wi = 0 / 1) of a K half
// takes the even / odd 16-row stages; the pair's tiles are added through LDS after the loop (fixed order: even + odd).
// QUART (round 6): K = 64 AND a 64-wide Cout tile (grid.y = Cout / 64): the GRU blocks' weight gradients of TSRN -- W_ih
// (64 -> 192), the 1x1 convolution (64 -> 64) and, as the diagonal blocks of the [192 x 64] cross product of the gate
// gradients with h_prev of both directions, W_hh.  Both operands' rows are 256 bytes, both keep the 512-byte row slots
// (lane quarters fetch the same 64 columns), all four waves own the SAME 64 x 64 tile and take every fourth 16-row
// stage; waves 1-3 are folded into wave 0 through LDS after the loop (fixed order).  These layers ran on the generic
// kernels at 67-99 us for 67-134 MB (profiles/r06c_c1_bygrid.txt).
template <bool HALF, bool QUART = false>
__global__ __launch_bounds__(256, 2) void linear_wgrad_stream_kernel(const float* __restrict__ X,
                                                                     const float* __restrict__ dY,
                                                                     float* __restrict__ PART, int M, int ldx, int ldd,
                                                                     int K, int Cout, int rows_per_split, int want_bias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lw_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  const int k0 = QUART ? 0 : blockIdx.x * 128, co0 = blockIdx.y * (QUART ? 64 : 128);
  const int row_beg = blockIdx.z * rows_per_split;
  const int row_end = min(M, row_beg + rows_per_split);
  const int nchunks = (row_end - row_beg) / LW_ROWS;          // the launcher hands out multiples of 16 rows
  float* slot = PART + (size_t)blockIdx.z * ((size_t)Cout * K + Cout);

  // DMA pieces of this wave: two of dY, two of X (piece = 2 rows x 128 columns = one 1 KB wave transfer)
  const float* gp[4];
  unsigned ldso[4];
  size_t gstep[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int P = wave + 4 * i, mat = P >> 3, q = P & 7;
    const int ld = mat ? ldx : ldd;
    gp[i] = (mat ? X + k0 : dY + co0) + (size_t)(row_beg + 2 * q + lh) * ld +
            4 * ((QUART || (HALF && !mat)) ? (li & 15) : li);
    gstep[i] = (size_t)LW_ROWS * ld;
    ldso[i] = mat * LW_MAT + q * LW_PIECE;
  }
  const unsigned lbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lw_smem;
  auto issue = [&](int c) {
    const unsigned st = (unsigned)(c % LW_NS) * LW_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp[i],
                                       (__attribute__((address_space(3))) void*)(lw_smem + st + ldso[i]), 16, 0, 0);
      gp[i] += gstep[i];
    }
  };

  lw_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum[2] = {0.f, 0.f};
  const bool do_bias = want_bias && blockIdx.x == 0 && (QUART || wj == 0);

  // fragment read addresses inside a stage: rows 8 lh + j, columns 64 w + 2 li (+ t)
  const unsigned aoff = lbase + (4 * lh) * LW_PIECE + (((HALF || QUART) ? 0 : 64 * wi) + 2 * li) * 4;
  const unsigned boff = lbase + LW_MAT + (4 * lh) * LW_PIECE + ((QUART ? 0 : 64 * wj) + 2 * li) * 4;

#pragma unroll
  for (int c = 0; c < LW_NS - 1; ++c)
    if (c < nchunks) issue(c);
  for (int c = 0; c < nchunks; ++c) {
    // stage c has landed once at most the transfers of the later stages are outstanding (vmcnt is in order)
    const int later = min(LW_NS - 2, nchunks - 1 - c);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // every wave's pieces of stage c are in LDS; stage c - 1 is free again
    if (c + LW_NS - 1 < nchunks) issue(c + LW_NS - 1);
    if (QUART) {
      if ((c & 3) != wave) continue;                          // (wave-uniform) every wave takes every fourth stage
    } else if (HALF && (c & 1) != wi) continue;               // (wave-uniform) the partner wave takes this stage
    const unsigned st = (unsigned)(c % LW_NS) * LW_STAGE;
    LwRaw raw;
    lw_read(raw, aoff + st, boff + st);
    lw_wait(raw);
    lw_bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      lw_split(raw.a, t, ah[t], al[t]);
      lw_split(raw.b, t, bh[t], bl[t]);
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bsum[0] += raw.a[j][0];
        bsum[1] += raw.a[j][1];
      }
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
  if (QUART) {
    // waves 1..3 -> LDS -> added into wave 0 in wave order: the stage buffers are idle now (3 x 16.9 KB of the 67.5 KB)
    __syncthreads();
    if (wave > 0) {
      float* red = reinterpret_cast<float*>(lw_smem) + (wave - 1) * (4 * 16 + 2) * 64 + lane;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((a * 2 + b) * 16 + r) * 64] = acc[a][b][r];
      red[64 * 64] = bsum[0];
      red[65 * 64] = bsum[1];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 0; w < 3; ++w) {
      const float* red = reinterpret_cast<const float*>(lw_smem) + w * (4 * 16 + 2) * 64 + lane;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * 2 + b) * 16 + r) * 64];
      bsum[0] += red[64 * 64];
      bsum[1] += red[65 * 64];
    }
  } else if (HALF) {
    // odd-stage tiles (wi = 1) -> LDS -> added to the even-stage tiles (wi = 0): the stage buffers are idle now
    __syncthreads();
    float* red = reinterpret_cast<float*>(lw_smem) + wj * (4 * 16 + 2) * 64 + lane;
    if (wi == 1) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((a * 2 + b) * 16 + r) * 64] = acc[a][b][r];
      red[64 * 64] = bsum[0];
      red[65 * 64] = bsum[1];
    }
    __syncthreads();
    if (wi == 1) return;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] += red[((a * 2 + b) * 16 + r) * 64];
    bsum[0] += red[64 * 64];
    bsum[1] += red[65 * 64];
  }
  // tile (a, b): row i of the MFMA is co = co0 + 64 wi + 2 i + a, column n is k = k0 + 64 wj + 2 n + b
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int co = co0 + ((HALF || QUART) ? 0 : 64 * wi) + 2 * i + a;
      float2 v = make_float2(acc[a][0][r], acc[a][1][r]);
      *reinterpret_cast<float2*>(slot + (size_t)co * K + k0 + (QUART ? 0 : 64 * wj) + 2 * li) = v;
    }
  if (do_bias) {
    bsum[0] += __shfl_xor(bsum[0], 32);
    bsum[1] += __shfl_xor(bsum[1], 32);
    if (lh == 0)
      *reinterpret_cast<float2*>(slot + (size_t)Cout * K + co0 + ((HALF || QUART) ? 0 : 64 * wi) + 2 * li) = make_float2(bsum[0], bsum[1]);
  }
}

// dst (+)= sum over slots, fixed order.  Block: 32 slot groups x 8 float4 elements; every thread keeps its group's
// loads in flight together, the groups are folded through LDS in group order.
#define LW_RG 32
__global__ __launch_bounds__(256) void slot_reduce_kernel(const float* __restrict__ PART, float* __restrict__ dw,
                                                          float* __restrict__ dbias, long n_dw4, long n_all4,
                                                          long slot_floats, int nslots, int accumulate) {
  __shared__ float4 red[LW_RG][8];
  const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + e;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n_all4) {
    const int per = (nslots + LW_RG - 1) / LW_RG;
    const int s0 = g * per, s1 = min(nslots, s0 + per);
    for (int b = s0; b < s1; b += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = (b + u < s1) ? reinterpret_cast<const float4*>(PART + (size_t)(b + u) * slot_floats)[i]
                            : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  red[g][e] = s;
  __syncthreads();
  if (threadIdx.x < 8 && i < n_all4) {
    float4 t = red[0][e];
#pragma unroll
    for (int q = 1; q < LW_RG; ++q) { t.x += red[q][e].x; t.y += red[q][e].y; t.z += red[q][e].z; t.w += red[q][e].w; }
    float* o = i < n_dw4 ? dw + i * 4 : (dbias ? dbias + (i - n_dw4) * 4 : nullptr);
    if (o) {
      if (accumulate) {
        float4 p = *reinterpret_cast<float4*>(o);
        t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w;
      }
      *reinterpret_cast<float4*>(o) = t;
    }
  }
}

#ifndef LW_BLOCKS
#define LW_BLOCKS 512
#endif
static inline bool lw_quart(int K, int Cout) { return K == 64 && Cout % 64 == 0 && Cout <= 256; }
static void lw_splits(long M, int K, int Cout, int& sp, int& rows) {
  const int tiles = lw_quart(K, Cout) ? Cout / 64 : (K / 128) * (Cout >= 128 ? Cout / 128 : 1);
  sp = LW_BLOCKS / tiles;
  if (sp < 1) sp = 1;
  rows = (int)(((M + sp - 1) / sp + LW_ROWS - 1) / LW_ROWS) * LW_ROWS;
  if (rows < 4 * LW_ROWS) rows = 4 * LW_ROWS;
  if (lw_quart(K, Cout) && rows < 16 * LW_ROWS) rows = 16 * LW_ROWS;      // four stages per wave at least
  sp = (int)((M + rows - 1) / rows);
}
int focr_linear_wgrad_eligible(long M, int K, int Cout, int ldx, int ldd) {
  static const bool half_ok = !(getenv("FOCR_LW_HALF") && getenv("FOCR_LW_HALF")[0] == '0');      // A/B: 128 -> 64 on the generic kernel
  static const bool quart_ok = !(getenv("FOCR_LW_QUART") && getenv("FOCR_LW_QUART")[0] == '0');  // A/B: K = 64 layers on the generic kernels
  if (lw_quart(K, Cout)) return quart_ok && M >= 4096 && M % LW_ROWS == 0 && ldx % 4 == 0 && ldd % 4 == 0;
  if (Cout == 64 && !half_ok) return 0;
  return M >= 1024 && M % LW_ROWS == 0 && K % 128 == 0 && (Cout % 128 == 0 || Cout == 64) && ldx % 4 == 0 && ldd % 4 == 0 &&
         (long)K * Cout <= 128 * 384;
}
long focr_linear_wgrad_ws_floats(long M, int K, int Cout) {
  int sp, rows;
  lw_splits(M, K, Cout, sp, rows);
  return (long)sp * ((long)Cout * K + Cout);
}
// x [M][ldx] (K columns used), dy [M][ldd] (Cout columns used) -> dw [Cout][K], dbias [Cout] (optional).
// accumulate != 0: add to what dw / dbias hold (buffers the caller zeroed once per step), else overwrite.
int focr_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, long M, int K,
                      int Cout, int ldx, int ldd, int accumulate, hipStream_t stream) {
  int sp, rows;
  lw_splits(M, K, Cout, sp, rows);
  const long slot = (long)Cout * K + Cout;
  if (!ws || ws_floats < (long)sp * slot) return 1;
  static focr_dev_flags attr_done;
  if (focr_dev_first(attr_done)) {
    if (hipFuncSetAttribute((const void*)linear_wgrad_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LW_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)linear_wgrad_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LW_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)linear_wgrad_stream_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            LW_LDS) != hipSuccess)
      return 1;                                        // caller falls back to the generic weight-gradient kernel
    focr_dev_mark(attr_done);
  }
  if (lw_quart(K, Cout))
    hipLaunchKernelGGL((linear_wgrad_stream_kernel<false, true>), dim3(1, Cout / 64, sp), 256, LW_LDS, stream, x, dy, ws, (int)M,
                       ldx, ldd, K, Cout, rows, dbias ? 1 : 0);
  else if (Cout == 64)
    hipLaunchKernelGGL(linear_wgrad_stream_kernel<true>, dim3(K / 128, 1, sp), 256, LW_LDS, stream, x, dy, ws, (int)M, ldx,
                       ldd, K, Cout, rows, dbias ? 1 : 0);
  else
    hipLaunchKernelGGL(linear_wgrad_stream_kernel<false>, dim3(K / 128, Cout / 128, sp), 256, LW_LDS, stream, x, dy, ws,
                       (int)M, ldx, ldd, K, Cout, rows, dbias ? 1 : 0);
  const long n_dw4 = (long)Cout * K / 4, n_all4 = n_dw4 + (dbias ? Cout / 4 : 0);
  hipLaunchKernelGGL(slot_reduce_kernel, dim3((int)((n_all4 + 7) / 8)), 256, 0, stream, (const float*)ws, dw, dbias,
                     n_dw4, n_all4, slot, sp, accumulate);
  return 0;
}

// Implicit-GEMM NHWC convolution / linear on the bf16 MFMA pipe with split operands ("bf16x3"):
// same tiling, gather and epilogue as conv_igemm.hip's VEC path (Cin % 32 == 0), but the LDS tiles
// hold bf16 hi/lo pairs (x = hi + lo) and each 16-deep k-step issues
//   A_hi.B_hi + A_hi.B_lo + A_lo.B_hi   on v_mfma_f32_32x32x16_bf16  (fp32 accumulate).
// End-to-end error equals plain fp32's (tools/exp_split_precision.py); MFMA time is 3/16 of the
// f32-MFMA kernel's, so these layers become L2/HBM-traffic bound instead.
#include "focr_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

#define XBM 128
#define XBK 32
#define XRP 40   // LDS row pitch in bf16 (32 + 8): 80 B, conflict-free ds_read_b128 fragments

struct ConvGeomX {
  int N, H, W, Cin, OH, OW, Cout, KH, KW, padH, padW, Ktot, M, ldy, ldr, ldx;
  int xcd_nk = 0, xcd_sp = 0;      // conv_wgrad_bx3_wide_kernel, XCD-aware 1-D grid: k tiles, pixel splits (multiple of 8); 0 = 3-D grid
};

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) { focr_split4(v, hi, lo); }

template <int NT>
__global__ __launch_bounds__(256) void conv_fwd_bx3_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ R, float* __restrict__ Y,
                                                           ConvGeomX g, float alpha, int relu,
                                                           float* __restrict__ PART, int chunks_per_split) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) __bf16 Ah[XBM * XRP], Al[XBM * XRP];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[BN * XRP], Bl[BN * XRP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * XBM, n0 = blockIdx.y * BN;
  // split-K (PART != nullptr): block z contracts the K chunks [cbeg, cend) and stores its raw tile into slot z
  const int cbeg = PART ? blockIdx.z * chunks_per_split : 0;
  const int nchunks = PART ? min(g.Ktot / XBK, cbeg + chunks_per_split) : g.Ktot / XBK;

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

  load_chunk(cbeg);
  for (int c = cbeg; c < nchunks; ++c) {
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
  if (PART) {
    float* slot = PART + (size_t)blockIdx.z * g.M * g.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = n0 + nt * 32 + li;
      if (co >= g.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (p < g.M) slot[(size_t)p * g.Cout + co] = acc[nt][r];
      }
    }
    return;
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

// y = epilogue(alpha * sum_z PART[z] + bias + residual): the K splits folded in a fixed order (deterministic)
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ PART,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ R, float* __restrict__ Y,
                                                                 long M, int Cout, int ldy, int ldr, int splits,
                                                                 float alpha, int relu) {
  const long n4 = M * Cout / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 s = reinterpret_cast<const float4*>(PART)[i];
    for (int z = 1; z < splits; ++z) {
      const float4 v = reinterpret_cast<const float4*>(PART + (size_t)z * M * Cout)[i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const long p = (i * 4) / Cout;
    const int co = (int)((i * 4) - p * Cout);
    float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = alpha * o[e] + (bias ? bias[co + e] : 0.f);
      if (R) v += R[(size_t)p * ldr + co + e];
      if (relu) v = fmaxf(v, 0.f);
      o[e] = v;
    }
    *reinterpret_cast<float4*>(Y + (size_t)p * ldy + co) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Layers with few output tiles and a long contraction (the STN head's 1x2 .. 2x8 px maps, the CRNN tail: 8 .. 208
// blocks walking 64-144 K chunks each) are split along K: every split writes its raw tile to a workspace slot and
// conv_splitk_reduce_kernel folds the slots in a fixed order and applies the epilogue -- deterministic, no atomics.
static int splitk_plan(int M, int Cout, int Ktot, int nt, int& cps) {
  const int blocks = ((M + XBM - 1) / XBM) * ((Cout + 32 * nt - 1) / (32 * nt));
  const int nchunks = Ktot / XBK;
  if (blocks >= 256 || nchunks < 16 || Cout % 4) return 1;
  int splits = (512 + blocks - 1) / blocks;             // aim at two blocks per CU
  if (splits > nchunks / 4) splits = nchunks / 4;       // at least four chunks per block
  if (splits > 32) splits = 32;
  if (splits < 2) return 1;
  cps = (nchunks + splits - 1) / splits;
  return (nchunks + cps - 1) / cps;
}
long focr_conv_fwd_bx3_ws_floats(int M, int Cin, int Cout, int Ktot) {
  int cps = 0;
  const int splits = splitk_plan(M, Cout, Ktot, Cout > 32 ? 2 : 1, cps);
  return splits > 1 ? (long)splits * M * Cout : 0;
}
// launcher used by focr_conv2d_fwd (conv_igemm.hip) when precision == bf16x3 and the layer is VEC-capable
int focr_conv_fwd_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y,
                      int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int padH, int padW,
                      int M, int ldy, int ldr, int ldx, float alpha, int relu, float* ws, long ws_floats,
                      hipStream_t stream) {
  ConvGeomX g{N, H, W, Cin, OH, OW, Cout, KH, KW, padH, padW, KH * KW * Cin, M, ldy, ldr, ldx};
  // widest column tile that does not waste MFMA work: the A (activation) tile is re-read once per column block
  int nt = Cout > 32 ? 2 : 1;   // a 128-column tile (NT=4) was measured: no gain, these GEMMs are latency bound
  int cps = 0;
  int splits = ws ? splitk_plan(M, Cout, g.Ktot, nt, cps) : 1;
  if (splits > 1 && ws_floats < (long)splits * M * Cout) splits = 1;
  float* part = splits > 1 ? ws : nullptr;
  dim3 grid((M + XBM - 1) / XBM, (Cout + 32 * nt - 1) / (32 * nt), splits);
  if (nt == 2)
    hipLaunchKernelGGL((conv_fwd_bx3_kernel<2>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu, part, cps);
  else
    hipLaunchKernelGGL((conv_fwd_bx3_kernel<1>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu, part, cps);
  if (part) {
    const long n4 = (long)M * Cout / 4;
    int rb = (int)((n4 + 255) / 256);
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(rb), 256, 0, stream, (const float*)part, bias, residual, y, (long)M,
                       Cout, ldy, ldr, splits, alpha, relu);
  }
  return 0;
}

// =======================================================================================
// wgrad on the bf16 pipe:  dW[co][k] += sum_p dY[p][co] * A[p][k],  k inside one tap (Cin % 64 == 0)
// The contraction index is the pixel, so both operands are staged TRANSPOSED in LDS
// (Dt[co][pixel], Xt[k][pixel], bf16 hi/lo, pitch 72 -> 16-byte aligned, conflict-free fragment reads);
// two adjacent pixels are packed per 32-bit LDS store.  64 co x 64 k tile, 4 waves as 2x2 of 32x32,
// 64 pixels per stage (4 MFMA k-steps x 3 split products), split over pixel ranges + fp32 atomics.
// =======================================================================================
#define WTP 72

// 4 pixels x 4 channels held by one thread -> for each channel one 8-byte store of its 4 pixels (hi and lo)
__device__ __forceinline__ void put_quad(__bf16* Th, __bf16* Tl, int c0, int px0, const float4 (&r)[4]) {
  const float v[4][4] = {{r[0].x, r[0].y, r[0].z, r[0].w}, {r[1].x, r[1].y, r[1].z, r[1].w},
                         {r[2].x, r[2].y, r[2].z, r[2].w}, {r[3].x, r[3].y, r[3].z, r[3].w}};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bf16x4 h, l;
    focr_split4(v[0][e], v[1][e], v[2][e], v[3][e], h, l);
    *reinterpret_cast<bf16x4*>(&Th[(c0 + e) * WTP + px0]) = h;
    *reinterpret_cast<bf16x4*>(&Tl[(c0 + e) * WTP + px0]) = l;
  }
}

__global__ __launch_bounds__(256) void conv_wgrad_bx3_kernel(const float* __restrict__ X, const float* __restrict__ dY,
                                                             float* __restrict__ dW, float* __restrict__ dbias,
                                                             ConvGeomX g, int ldd, int pix_per_split) {
  __shared__ __attribute__((aligned(16))) __bf16 Dth[64 * WTP], Dtl[64 * WTP], Xth[64 * WTP], Xtl[64 * WTP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int k0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int pbeg = blockIdx.z * pix_per_split;
  const int pend = min(g.M, pbeg + pix_per_split);
  if (pbeg >= pend) return;
  const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin;
  const int tkh = tap / g.KW, tkw = tap - tkh * g.KW;
  // staging: the 16 lanes of a group own 16 consecutive pixel quads (conflict-free 8-byte LDS stores),
  // the group index selects the 4-channel column
  const int pq = tid & 15, c4 = (tid >> 4) * 4;
  const bool do_bias = dbias != nullptr && blockIdx.x == 0;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool co_full = co0 + c4 + 3 < g.Cout;

  float4 dv[4], xv[4];
  auto load_chunk = [&](int pc) {
    int p = pc + 4 * pq;
    int n = 0, oy = 0, ox = 0;
    if (p < pend) {                       // decode the first pixel once, then walk
      n = p / (g.OH * g.OW);
      int rem = p - n * (g.OH * g.OW);
      oy = rem / g.OW;
      ox = rem - oy * g.OW;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), x = d;
      if (p + j < pend) {
        if (co_full) {
          d = *reinterpret_cast<const float4*>(dY + (size_t)(p + j) * ldd + co0 + c4);
        } else {
          float t[4] = {0.f, 0.f, 0.f, 0.f};
          for (int e = 0; e < 4; ++e)
            if (co0 + c4 + e < g.Cout) t[e] = dY[(size_t)(p + j) * ldd + co0 + c4 + e];
          d = make_float4(t[0], t[1], t[2], t[3]);
        }
        int iy = oy - g.padH + tkh, ix = ox - g.padW + tkw;
        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
          x = *reinterpret_cast<const float4*>(X + ((size_t)((n * g.H + iy) * g.W + ix) * g.ldx + ci0 + c4));
      }
      dv[j] = d;
      xv[j] = x;
      if (++ox == g.OW) {
        ox = 0;
        if (++oy == g.OH) { oy = 0; ++n; }
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int wi = wave >> 1, wj = wave & 1;
  const int aoff = (wi * 32 + li) * WTP + 8 * lh, boff = (wj * 32 + li) * WTP + 8 * lh;

  load_chunk(pbeg);
  for (int pc = pbeg; pc < pend; pc += 64) {
    put_quad(Dth, Dtl, c4, 4 * pq, dv);
    put_quad(Xth, Xtl, c4, 4 * pq, xv);
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bsum[0] += dv[j].x; bsum[1] += dv[j].y; bsum[2] += dv[j].z; bsum[3] += dv[j].w;
      }
    }
    __syncthreads();
    if (pc + 64 < pend) load_chunk(pc + 64);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Dth[aoff + 16 * m]);
      bf16x8 al = *reinterpret_cast<const bf16x8*>(&Dtl[aoff + 16 * m]);
      bf16x8 bh = *reinterpret_cast<const bf16x8*>(&Xth[boff + 16 * m]);
      bf16x8 bl = *reinterpret_cast<const bf16x8*>(&Xtl[boff + 16 * m]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_bias) {   // the 16 lanes sharing (tid >> 4) own the same 4 columns: fold them in LDS, 64 atomics per block
    float* red = reinterpret_cast<float*>(Dth);          // 16 x 64 floats (4096 B) fit in the 9216-B tile
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pq * 64 + c4 + e] = bsum[e];
    __syncthreads();
    if (tid < 64 && co0 + tid < g.Cout) {
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s2 += red[r * 64 + tid];
      atomicAdd(&dbias[co0 + tid], s2);
    }
  }
  int k = k0 + wj * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (co < g.Cout) atomicAdd(&dW[(size_t)co * g.Ktot + k], acc[r]);
  }
}

// 128 co x 128 k tile (4 waves x (64 x 64) = 4 accumulator tiles each): every dY / X element of the pixel range is
// fetched ONCE (the 64 x 64 kernel re-reads each operand once per tile of the other), twice the MFMA work per barrier
// and a quarter of the atomics per pixel.  For layers with Cin % 128 == 0 and Cout % 128 == 0 (the transformer
// linears): the k block stays inside one tap.
// PART != nullptr: split z writes its tile into its own full-size slot PART[z][Cout][Ktot] with plain stores and
// partial_sum_kernel folds the slots (no same-address atomics, which cost as much as the rest of this kernel).
__global__ __launch_bounds__(256) void conv_wgrad_bx3_wide_kernel(const float* __restrict__ X,
                                                                  const float* __restrict__ dY, float* __restrict__ dW,
                                                                  float* __restrict__ dbias, float* __restrict__ PART,
                                                                  ConvGeomX g, int ldd, int pix_per_split) {
  __shared__ __attribute__((aligned(16))) __bf16 Dth[128 * WTP], Dtl[128 * WTP], Xth[128 * WTP], Xtl[128 * WTP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (g.xcd_nk > 0) {
    // XCD-aware decode of a 1-D grid: blocks are dealt round-robin to the 8 XCDs (id % 8), each with its own L2.  The pixel
    // split is tied to the XCD, so an XCD's blocks -- all k / co tiles of ONE eighth of the pixels -- re-read a working set
    // (that eighth of X and dY) that fits its L2 instead of streaming both whole tensors through every L2
    const int id = blockIdx.x, xcd = id & 7, t = id >> 3, spx = g.xcd_sp >> 3;
    bz = xcd + 8 * (t % spx);
    const int tile = t / spx;
    bx = tile % g.xcd_nk;
    by = tile / g.xcd_nk;
  }
  const int k0 = bx * 128, co0 = by * 128;
  const int pbeg = bz * pix_per_split;
  const int pend = min(g.M, pbeg + pix_per_split);
  if (pbeg >= pend) return;
  const int tap = k0 / g.Cin, ci0 = k0 - tap * g.Cin;
  const int tkh = tap / g.KW, tkw = tap - tkh * g.KW;
  const int pq = tid & 15, c4 = (tid >> 4) * 4;
  const bool do_bias = dbias != nullptr && bx == 0;
  float bsum[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  float4 dv0[4], dv1[4], xv0[4], xv1[4];
  auto load_chunk = [&](int pc) {
    int p = pc + 4 * pq;
    int n = 0, oy = 0, ox = 0;
    if (p < pend) {
      n = p / (g.OH * g.OW);
      int rem = p - n * (g.OH * g.OW);
      oy = rem / g.OW;
      ox = rem - oy * g.OW;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 d0 = z, d1 = z, x0 = z, x1 = z;
      if (p + j < pend) {
        const float* dp = dY + (size_t)(p + j) * ldd + co0 + c4;
        d0 = *reinterpret_cast<const float4*>(dp);
        d1 = *reinterpret_cast<const float4*>(dp + 64);
        int iy = oy - g.padH + tkh, ix = ox - g.padW + tkw;
        if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) {
          const float* xp = X + ((size_t)((n * g.H + iy) * g.W + ix) * g.ldx + ci0 + c4);
          x0 = *reinterpret_cast<const float4*>(xp);
          x1 = *reinterpret_cast<const float4*>(xp + 64);
        }
      }
      dv0[j] = d0; dv1[j] = d1; xv0[j] = x0; xv1[j] = x1;
      if (++ox == g.OW) {
        ox = 0;
        if (++oy == g.OH) { oy = 0; ++n; }
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int wi = wave >> 1, wj = wave & 1;
  const int aoff = (wi * 64 + li) * WTP + 8 * lh, boff = (wj * 64 + li) * WTP + 8 * lh;

  load_chunk(pbeg);
  for (int pc = pbeg; pc < pend; pc += 64) {
    put_quad(Dth, Dtl, c4, 4 * pq, dv0);
    put_quad(Dth, Dtl, c4 + 64, 4 * pq, dv1);
    put_quad(Xth, Xtl, c4, 4 * pq, xv0);
    put_quad(Xth, Xtl, c4 + 64, 4 * pq, xv1);
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bsum[0][0] += dv0[j].x; bsum[0][1] += dv0[j].y; bsum[0][2] += dv0[j].z; bsum[0][3] += dv0[j].w;
        bsum[1][0] += dv1[j].x; bsum[1][1] += dv1[j].y; bsum[1][2] += dv1[j].z; bsum[1][3] += dv1[j].w;
      }
    }
    __syncthreads();
#if defined(WGW_ABL) && (WGW_ABL & 1)
    if (pc + 64 < pend && pc < pbeg + 64) load_chunk(pc + 64);
#else
    if (pc + 64 < pend) load_chunk(pc + 64);
#endif
#if defined(WGW_ABL) && (WGW_ABL & 2)
    if (pc == pend + 12345)
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(&Dth[aoff + t * 32 * WTP + 16 * m]);
        al[t] = *reinterpret_cast<const bf16x8*>(&Dtl[aoff + t * 32 * WTP + 16 * m]);
        bh[t] = *reinterpret_cast<const bf16x8*>(&Xth[boff + t * 32 * WTP + 16 * m]);
        bl[t] = *reinterpret_cast<const bf16x8*>(&Xtl[boff + t * 32 * WTP + 16 * m]);
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
    __syncthreads();
  }
  if (do_bias) {   // fold the 16 pixel-quad lanes of each column in LDS: 128 atomics per block
    float* red = reinterpret_cast<float*>(Dth);          // 16 x 128 floats = 8 KB of the 18 KB tile
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[pq * 128 + c4 + e] = bsum[0][e];
      red[pq * 128 + 64 + c4 + e] = bsum[1][e];
    }
    __syncthreads();
    if (tid < 128) {
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s2 += red[r * 128 + tid];
      atomicAdd(&dbias[co0 + tid], s2);
    }
  }
  float* out = PART ? PART + (size_t)bz * g.Cout * g.Ktot : dW;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      int k = k0 + wj * 64 + b * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int co = co0 + wi * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (PART) out[(size_t)co * g.Ktot + k] = acc[a][b][r];
        else atomicAdd(&out[(size_t)co * g.Ktot + k], acc[a][b][r]);
      }
    }
}

// dst[i] += sum_s PART[s][i], i < n4 float4; grid.y groups of slots, one atomic per element and group
#define PS_RG 8
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ PART, float* __restrict__ dst,
                                                          long n4, int nslots) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int s0 = (int)((long)nslots * blockIdx.y / PS_RG), s1 = (int)((long)nslots * (blockIdx.y + 1) / PS_RG);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int b = s0; b < s1; ++b) {
    float4 v = reinterpret_cast<const float4*>(PART)[(size_t)b * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* o = dst + i * 4;
  atomicAdd(o, s.x);
  atomicAdd(o + 1, s.y);
  atomicAdd(o + 2, s.z);
  atomicAdd(o + 3, s.w);
}

#ifndef WGRADW_BLOCKS
#define WGRADW_BLOCKS 512
#endif

static void wide_splits(int M, int Ktot, int Cout, int& sp, int& pp) {
  const int tiles = (Ktot / 128) * (Cout / 128);
  sp = WGRADW_BLOCKS / tiles;
  const int maxsp = (M + 511) / 512;                 // >= 8 reduction chunks per block
  if (sp > maxsp) sp = maxsp;
  if (sp < 1) sp = 1;
  pp = (((M + sp - 1) / sp + 63) / 64) * 64;
  sp = (M + pp - 1) / pp;
}
// workspace floats of the partial-slot path of the 128 x 128 kernel (0: not applicable / too large to be worth it)
long focr_conv_wgrad_bx3_ws_floats(int M, int Cin, int Cout, int Ktot) {
  if (!(Cin % 128 == 0 && Cout % 128 == 0)) return 0;
  int sp, pp;
  wide_splits(M, Ktot, Cout, sp, pp);
  const long n = (long)sp * Cout * Ktot;
  // small weight matrices only (transformer linears): with MB-sized slots the extra write + read costs more than the
  // atomics (measured on the 512 x 2304 CRNN layer: 162 -> 275 us)
  return (sp > 1 && (long)Cout * Ktot <= 65536) ? n : 0;
}

int focr_conv_wgrad_bx3(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                        int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int padH, int padW, int M,
                        int ldd, int ldx, int splits, int pps, hipStream_t stream) {
  ConvGeomX g{N, H, W, Cin, OH, OW, Cout, KH, KW, padH, padW, KH * KW * Cin, M, Cout, Cout, ldx};
  if (Cin % 128 == 0 && Cout % 128 == 0) {
    int sp, pp;
    wide_splits(M, g.Ktot, Cout, sp, pp);
    const long need = focr_conv_wgrad_bx3_ws_floats(M, Cin, Cout, g.Ktot);
    float* part = (ws && need > 0 && ws_floats >= need) ? ws : nullptr;
    dim3 gridw(g.Ktot / 128, Cout / 128, sp);
    // large layers (the SLD / text-focus ResNets: X and dY are 16 MB each and every tile re-reads them): eight pixel splits,
    // one per XCD (see the kernel); needs >= 8 chunks of 64 pixels per split and no partial-slot path
    static const int xcd_mode = getenv("FOCR_WGW_XCD") ? atoi(getenv("FOCR_WGW_XCD")) : 1;
    if (xcd_mode && !part && (long)g.Ktot * Cout >= (1l << 22) && M >= 8 * 512) {   // (512 -> 512 is slower this way: 190 -> 212 us)
      const int spx = 8 * xcd_mode;
      pp = (((M + spx - 1) / spx + 63) / 64) * 64;
      if ((long)pp * (spx - 1) < M) {
        g.xcd_nk = g.Ktot / 128;
        g.xcd_sp = spx;
        gridw = dim3((g.Ktot / 128) * (Cout / 128) * spx, 1, 1);
      }
    }
    hipLaunchKernelGGL(conv_wgrad_bx3_wide_kernel, gridw, 256, 0, stream, x, dy, dw, dbias, part, g, ldd, pp);
    if (part) {
      const long n4 = (long)Cout * g.Ktot / 4;
      hipLaunchKernelGGL(partial_sum_kernel, dim3((int)((n4 + 255) / 256), PS_RG), 256, 0, stream, part, dw, n4, sp);
    }
    return 0;
  }
  dim3 grid(g.Ktot / 64, (Cout + 63) / 64, splits);
  hipLaunchKernelGGL(conv_wgrad_bx3_kernel, grid, 256, 0, stream, x, dy, dw, dbias, g, ldd, pps);
  return 0;
}

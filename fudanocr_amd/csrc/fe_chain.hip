// Row-local chains of the TBSRN FeatureEnhancer (reference tbsrn.py:76-92, 23-36, 153-163): everything between the
// attention output and the block's 128 -> 64 projection is independent per token row, so each chain runs in ONE
// kernel with the row held in registers from its first GEMM to its last store:
//
//   forward   fe_fwd_a : ctx -> O-proj + b (+ tok) -> std-LayerNorm1 -> [xhat1, rinv1] -> a1 xhat1 + b1 -> w_1 + b ->
//                        relu -> dropout -> [h]
//             fe_fwd_b : h -> w_2 + b (+ a1 xhat1 + b1) -> std-LayerNorm3 -> [xhat2, rinv2] -> a3 xhat2 + b3 ->
//                        linear 128 -> 64 + b (+ block input) -> [out]
//   backward  fe_bwd_a : d_out -> linear^T -> LayerNorm3 backward -> [d_s2] -> w_2^T -> relu/dropout mask (h) -> [d_hpre]
//             fe_bwd_b : d_hpre -> w_1^T (+ d_s2) -> LayerNorm1 backward -> [d_s1] -> O-proj^T -> [d_ctx]
//             fe_bwd_qkv : dqkv [rows, 384] -> packed QKV projection^T restricted to the 64 feature columns of the token
//                        (the positional-encoding half of the token has no gradient consumer) (+ d_s1[:, :64]) -> [d_feat]
//
// Before: each arrow group above was 2-3 separate launches with a 67 MB fp32 round trip between them (O-proj, LayerNorm,
// w_1, w_2, LayerNorm, linear; backward: six GEMM / LayerNorm passes): 14 resp. 22.5 row-matrix transfers per block; now 8
// resp. 13.5.  The normalised rows xhat are what is kept for the backward (not the affine outputs): the LayerNorm backward
// needs xhat and 1/(std+eps) only, and the affine output a xhat + b is rebuilt on load by its consumers.  The LayerNorm
// parameter gradients and the weight gradients of the linears that consume a LayerNorm output come out of ONE weight-
// gradient GEMM on xhat (fe_ln_lin_finish_kernel, algebra in its comment): no column reductions inside these kernels.
//
// Arithmetic: split-bf16 ("bf16x3": hi/lo operands, three products, fp32 accumulate) on v_mfma_f32_32x32x16_bf16, as
// every other contraction of the path.
//
// Lane layout.  A wave owns 32 token rows.  All GEMMs are evaluated as Y^T = W X^T: the weight rows are the MFMA's A
// operand (from LDS), the token rows its B operand, so lane (li = lane & 31, lh = lane >> 5) holds of row li the output
// columns 32 j + 8 g + 4 lh + e in register 4 g + e of accumulator tile j.  The contraction order of an MFMA is free, so
// k-slot (step s, half lh, element e) of every operand is DEFINED as column 16 s + 4 lh + (e & 3) + 8 (e >> 2): a lane's
// B fragment of step s = 2 j + m is then its own accumulator registers 8 m .. 8 m + 7 of tile j -- the output of one GEMM
// is the input operand of the next without any cross-lane movement, row statistics are in-lane sums plus one exchange
// with lane ^ 32, and rows in HBM are read / written as 16-byte pieces at columns 32 j + 8 g + 4 lh.  The weights are
// split to bf16 hi/lo and permuted to that k order once per block while they are staged into LDS (1 block of 8 waves
// per CU; 102 - 139 KB of weights resident).
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 fc_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 fc_bf16x4;

#define FC_D 128
#define FC_THREADS 512

namespace {

__device__ __forceinline__ int fc_perm(int c) {            // column -> k position inside a weight row in LDS
  const int w = c & 15;
  return (c & ~15) + 8 * ((w >> 2) & 1) + (w & 3) + 4 * (w >> 3);
}

// ---- weight staging ---------------------------------------------------------------------------------------------
// Round 5: every staging routine is TWO-PHASE -- all of a thread's global loads of ALL matrices and vectors of the
// prologue are issued first (compile-time trip counts, straight-line code), then converted and stored.  The one-pass
// loops they replace (load -> split -> LDS store per iteration, one iteration per L2 round trip, a branch per vector)
// made the prologue 12-39 us of EVERY launch: the chains measured 22-39 us on 2 048 rows and 53-103 us on 131 072
// (profiles/r05_fe_prologue.txt), i.e. their streaming part already ran at 5.5-6 TB/s and a fifth to a third of each
// launch was this prologue.
// Wg [N][K] row-major fp32 -> LDS rows n (pitch K + 8 bf16), k permuted, hi / lo planes.  4 consecutive columns map to
// 4 consecutive k positions (fc_perm), so every thread converts a float4 and stores two 8-byte vectors.
template <int K, int N, int THREADS = FC_THREADS>
struct FcStageW {
  static constexpr int KP = K + 8, Q = K / 4, IT = N * Q / THREADS;
  static_assert(N * Q % THREADS == 0, "staging trip count");
  float4 v[IT];
  __device__ __forceinline__ void load(const float* __restrict__ Wg) {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = threadIdx.x + j * THREADS, n = i / Q, c = 4 * (i - n * Q);
      v[j] = *reinterpret_cast<const float4*>(Wg + (size_t)n * K + c);
    }
  }
  __device__ __forceinline__ void store(__bf16* Wh, __bf16* Wl) const {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      const int i = threadIdx.x + j * THREADS, n = i / Q, c = 4 * (i - n * Q);
      fc_bf16x4 h, l;
      focr_split4(v[j], h, l);
      const int kk = fc_perm(c);
      *reinterpret_cast<fc_bf16x4*>(&Wh[n * KP + kk]) = h;
      *reinterpret_cast<fc_bf16x4*>(&Wl[n * KP + kk]) = l;
    }
  }
};
// Transposed: Wg [N][ldw] (columns 0 .. C - 1 used) -> LDS rows c (C rows, pitch N + 8), k = permuted n.  The data
// gradient y = dy W contracts over the weight's OUTPUT index.
// A thread owns 4 x 4 blocks (rows 4 n4 .. + 3, columns 4 c4 .. + 3): four float4 loads, transposed in registers, and --
// fc_perm keeps four consecutive n together -- ONE 8-byte store per column and plane.  Lanes run over n4, so the 32
// stores of a half-wave fill one LDS row contiguously (no bank conflicts); the element-wise form it replaces (eight
// 2-byte stores per float4, lanes four LDS rows apart = 8-way conflicts) made the backward chains' prologue 6-17 us
// longer than the forward chains' (profiles/r05_fe_prologue.txt).
template <int N, int C>
struct FcStageWt {
  static constexpr int KP = N + 8, N4 = N / 4, C4 = C / 4, IT = N4 * C4 / FC_THREADS;
  static_assert(N4 * C4 % FC_THREADS == 0, "staging trip count");
  float4 v[IT][4];
  __device__ __forceinline__ static void item(int it, int& n4, int& c4) {
    n4 = it % N4;
    c4 = it / N4;
  }
  __device__ __forceinline__ void load(const float* __restrict__ Wg, int ldw) {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      int n4, c4;
      item(threadIdx.x + j * FC_THREADS, n4, c4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[j][r] = *reinterpret_cast<const float4*>(Wg + (size_t)(4 * n4 + r) * ldw + 4 * c4);
    }
  }
  __device__ __forceinline__ void store(__bf16* Wh, __bf16* Wl) const {
#pragma unroll
    for (int j = 0; j < IT; ++j) {
      int n4, c4;
      item(threadIdx.x + j * FC_THREADS, n4, c4);
      const int kk = fc_perm(4 * n4);
      const float a[4][4] = {{v[j][0].x, v[j][0].y, v[j][0].z, v[j][0].w}, {v[j][1].x, v[j][1].y, v[j][1].z, v[j][1].w},
                             {v[j][2].x, v[j][2].y, v[j][2].z, v[j][2].w}, {v[j][3].x, v[j][3].y, v[j][3].z, v[j][3].w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        fc_bf16x4 h, l;
        focr_split4(a[0][e], a[1][e], a[2][e], a[3][e], h, l);
        *reinterpret_cast<fc_bf16x4*>(&Wh[(4 * c4 + e) * KP + kk]) = h;
        *reinterpret_cast<fc_bf16x4*>(&Wl[(4 * c4 + e) * KP + kk]) = l;
      }
    }
  }
};
// n <= THREADS floats of a bias / LayerNorm vector: one unconditional (clamped) load per thread, stored under a predicate
template <int THREADS = FC_THREADS>
__device__ __forceinline__ float fc_vec_load(const float* __restrict__ g, int n) {
  return g ? g[min((int)threadIdx.x, n - 1)] : 0.f;
}
__device__ __forceinline__ void fc_vec_store(float* s, int n, float x) {
  if ((int)threadIdx.x < n) s[threadIdx.x] = x;
}

// ---- rows in the lane layout ------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void fc_load_row(const float* __restrict__ p, int lh, f32x16 (&v)[NT]) {
#ifdef FC_TLAYOUT_PROBE      // timing probe only (wrong data): 128-column rows addressed as [tile][16-byte chunk][row], i.e.
                             // every wave access two 512-byte runs.  Upper bound of what a tile-transposed layout of the chain-
                             // internal tensors could buy (r04: fwd_a 68 -> 61, fwd_b 67 -> 65, bwd_a 80 -> 72, bwd_b 104 -> 99 us,
                             // the two QKV kernels slower): ~0.1 ms per step at best -- not pursued
  if (NT == 4) {
    const int li_ = threadIdx.x & 31;
    const float* tb = p - li_ * FC_D;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 t = *reinterpret_cast<const float4*>(tb + (8 * j + 2 * g + lh) * 128 + li_ * 4);
        v[j][4 * g] = t.x; v[j][4 * g + 1] = t.y; v[j][4 * g + 2] = t.z; v[j][4 * g + 3] = t.w;
      }
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(p + 32 * j + 8 * g + 4 * lh);
      v[j][4 * g] = t.x; v[j][4 * g + 1] = t.y; v[j][4 * g + 2] = t.z; v[j][4 * g + 3] = t.w;
    }
}
template <int NT>
__device__ __forceinline__ void fc_store_row(float* __restrict__ p, int lh, const f32x16 (&v)[NT]) {
#ifdef FC_TLAYOUT_PROBE
  if (NT == 4) {
    const int li_ = threadIdx.x & 31;
    float* tb = p - li_ * FC_D;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(tb + (8 * j + 2 * g + lh) * 128 + li_ * 4) =
            make_float4(v[j][4 * g], v[j][4 * g + 1], v[j][4 * g + 2], v[j][4 * g + 3]);
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(p + 32 * j + 8 * g + 4 * lh) =
          make_float4(v[j][4 * g], v[j][4 * g + 1], v[j][4 * g + 2], v[j][4 * g + 3]);
}
// the row pre-split for the attention kernels: 256 bf16 per row, every 4 columns as [hi4 | lo4] (x * mul = hi + lo): one
// 16-byte store per register group, exactly like the fp32 row
template <int NT>
__device__ __forceinline__ void fc_store_planes(__bf16* __restrict__ p, int lh, const f32x16 (&v)[NT], float mul) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      fc_bf16x4 h, l;
      focr_split4(v[j][4 * g] * mul, v[j][4 * g + 1] * mul, v[j][4 * g + 2] * mul, v[j][4 * g + 3] * mul, h, l);
      *reinterpret_cast<fc_bf16x8*>(p + 2 * (32 * j + 8 * g + 4 * lh)) = __builtin_shufflevector(h, l, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
// a vector over the columns (bias, LayerNorm a / b) from LDS, in the lane layout (two distinct addresses per read)
template <int NT>
__device__ __forceinline__ void fc_load_vec(const float* s, int lh, f32x16 (&v)[NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 t = *reinterpret_cast<const float4*>(s + 32 * j + 8 * g + 4 * lh);
      v[j][4 * g] = t.x; v[j][4 * g + 1] = t.y; v[j][4 * g + 2] = t.z; v[j][4 * g + 3] = t.w;
    }
}
// accumulator tiles -> split MFMA operand fragments of the next GEMM (k-step 2 j + m = registers 8 m .. 8 m + 7 of tile j)
template <int NT>
__device__ __forceinline__ void fc_frags(const f32x16 (&v)[NT], fc_bf16x8 (&h)[2 * NT], fc_bf16x8 (&l)[2 * NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float x[8] = {v[j][8 * m],     v[j][8 * m + 1], v[j][8 * m + 2], v[j][8 * m + 3],
                          v[j][8 * m + 4], v[j][8 * m + 5], v[j][8 * m + 6], v[j][8 * m + 7]};
      focr_split8(x, h[2 * j + m], l[2 * j + m]);
    }
}
// acc[j] += W[32 j + i][:] . x   (NTO output tiles, KS k-steps; W from LDS at pitch KP, first k position koff)
template <int NTO, int KS, int KP>
__device__ __forceinline__ void fc_gemm(const __bf16* Wh, const __bf16* Wl, int koff, int li, int lh,
                                        const fc_bf16x8 (&xh)[KS], const fc_bf16x8 (&xl)[KS], f32x16 (&acc)[NTO]) {
#pragma unroll
  for (int j = 0; j < NTO; ++j)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int o = (32 * j + li) * KP + koff + 16 * s + 8 * lh;
      const fc_bf16x8 ah = *reinterpret_cast<const fc_bf16x8*>(&Wh[o]);
      const fc_bf16x8 al = *reinterpret_cast<const fc_bf16x8*>(&Wl[o]);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xh[s], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, xl[s], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, xh[s], acc[j], 0, 0, 0);
    }
}
__device__ __forceinline__ float fc_rowsum(float v) { return v + __shfl_xor(v, 32, 64); }

// the reference's LayerNorm (tbsrn.py:33-36): (x - mean) / (std_unbiased + eps); v <- xhat, returns 1 / (std + eps)
__device__ __forceinline__ float fc_ln_fwd(f32x16 (&v)[4], float eps) {
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[j][r];
  const float mean = fc_rowsum(s) * (1.f / FC_D);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v[j][r] -= mean;
      q += v[j][r] * v[j][r];
    }
  const float sd = sqrtf(fc_rowsum(q) * (1.f / (FC_D - 1)));
  const float rinv = 1.f / (sd + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[j][r] *= rinv;
  return rinv;
}
// LayerNorm backward from xhat: d <- rinv (g - mean g) - (sum g xhat) xhat / ((D - 1) sd), g = a d   (ln_bwd_kernel's
// formula with u = xhat / rinv)
__device__ __forceinline__ void fc_ln_bwd(f32x16 (&d)[4], const f32x16 (&xh)[4], const float* a_lds, int lh, float rinv,
                                          float eps) {
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 av = *reinterpret_cast<const float4*>(a_lds + 32 * j + 8 * g + 4 * lh);
      const float a4[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = d[j][4 * g + e] * a4[e];
        d[j][4 * g + e] = t;
        sg += t;
        sgx += t * xh[j][4 * g + e];
      }
    }
  sg = fc_rowsum(sg);
  sgx = fc_rowsum(sgx);
  const float sd = 1.f / rinv - eps;
  const float k = sgx / ((FC_D - 1) * sd);
  const float mg = sg * (1.f / FC_D);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) d[j][r] = rinv * (d[j][r] - mg) - k * xh[j][r];
}
// v <- a v + b (LayerNorm affine part) with a, b in LDS
__device__ __forceinline__ void fc_affine(f32x16 (&v)[4], const float* a_lds, const float* b_lds, int lh) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 av = *reinterpret_cast<const float4*>(a_lds + 32 * j + 8 * g + 4 * lh);
      const float4 bv = *reinterpret_cast<const float4*>(b_lds + 32 * j + 8 * g + 4 * lh);
      v[j][4 * g] = av.x * v[j][4 * g] + bv.x;
      v[j][4 * g + 1] = av.y * v[j][4 * g + 1] + bv.y;
      v[j][4 * g + 2] = av.z * v[j][4 * g + 2] + bv.z;
      v[j][4 * g + 3] = av.w * v[j][4 * g + 3] + bv.w;
    }
}
// An opaque zero: added to an LDS base pointer it keeps the (tile-invariant) weight fragment reads inside the tile loop
// (hoisted, they would need > 500 VGPRs; linear_stream.hip uses the same device).
__device__ __forceinline__ int fc_opaque_zero() {
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  return z;
}

constexpr int KP128 = FC_D + 8;      // 136
constexpr int KP64 = 64 + 8;         // 72
constexpr int KP384 = 384 + 8;       // 392
constexpr int WSZ128 = FC_D * KP128; // bf16 elements of one plane of a 128 x 128 weight

}  // namespace

// =====================================================================================================================
// forward A: ctx -> O-proj (+ tok) -> LN1 -> [xhat1, rinv1] -> affine -> w_1 -> relu -> dropout -> [h]
// =====================================================================================================================
__global__ __launch_bounds__(FC_THREADS, 1) void fe_fwd_a_kernel(
    const float* __restrict__ ctx, const float* __restrict__ tok, const float* __restrict__ Wo,
    const float* __restrict__ bo, const float* __restrict__ a1, const float* __restrict__ b1,
    const float* __restrict__ W1, const float* __restrict__ bb1, float* __restrict__ xhat1, float* __restrict__ rinv1,
    float* __restrict__ hbuf, int ntiles, float eps, uint32_t drop_k, float drop_scale, uint32_t drop_seed,
    const uint64_t* __restrict__ epoch) {
  drop_seed = focr_epoch_seed32(drop_seed, epoch);
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* Woh = reinterpret_cast<__bf16*>(fc_smem);
  __bf16* Wol = Woh + WSZ128;
  __bf16* W1h = Wol + WSZ128;
  __bf16* W1l = W1h + WSZ128;
  float* vbo = reinterpret_cast<float*>(W1l + WSZ128);
  float* va1 = vbo + FC_D;
  float* vb1 = va1 + FC_D;
  float* vbb1 = vb1 + FC_D;
  {
    FcStageW<FC_D, FC_D> s0, s1;
    s0.load(Wo);
    s1.load(W1);
    const float x0 = fc_vec_load(bo, FC_D), x1 = fc_vec_load(a1, FC_D), x2 = fc_vec_load(b1, FC_D), x3 = fc_vec_load(bb1, FC_D);
    s0.store(Woh, Wol);
    s1.store(W1h, W1l);
    fc_vec_store(vbo, FC_D, x0);
    fc_vec_store(va1, FC_D, x1);
    fc_vec_store(vb1, FC_D, x2);
    fc_vec_store(vbb1, FC_D, x3);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 v[4], res[4];
    fc_load_row<4>(ctx + row * FC_D, lh, v);
    fc_load_row<4>(tok + row * FC_D, lh, res);
    fc_bf16x8 xh[8], xl[8];
    fc_frags<4>(v, xh, xl);
    const int z = fc_opaque_zero();
    fc_load_vec<4>(vbo + z, lh, v);                       // accumulators start at the bias
    fc_gemm<4, 8, KP128>(Woh + z, Wol + z, 0, li, lh, xh, xl, v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[j][r] += res[j][r];
    const float rinv = fc_ln_fwd(v, eps);
    fc_store_row<4>(xhat1 + row * FC_D, lh, v);
    if (lh == 0) rinv1[row] = rinv;
    fc_affine(v, va1 + z, vb1 + z, lh);
    fc_frags<4>(v, xh, xl);
    fc_load_vec<4>(vbb1 + z, lh, v);
    fc_gemm<4, 8, KP128>(W1h + z, W1l + z, 0, li, lh, xh, xl, v);
    // Dropout(relu(.)) (tbsrn.py:162-163): 64 Bernoulli keep bits per lane, bit-sliced from a xorshift stream
    // (P(keep) = drop_k / 65536, as in linear_stream.hip).  The backward needs no mask: a dropped element IS a zero of h.
    uint32_t keep[2] = {0xffffffffu, 0xffffffffu};
    if (drop_k) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        uint32_t x = hash32(drop_seed ^ hash32((uint32_t)(2 * t + u) * 0x9E3779B1U) ^ (uint32_t)lane * 0x85EBCA6BU) | 1u;
        uint32_t k = 0u;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
          x ^= x << 13; x ^= x >> 17; x ^= x << 5;
          k = ((drop_k >> b) & 1u) ? (k | x) : (k & x);
        }
        keep[u] = k;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float y = fmaxf(v[j][r], 0.f);
        if (drop_k) y = ((keep[j >> 1] >> (16 * (j & 1) + r)) & 1u) ? y * drop_scale : 0.f;
        v[j][r] = y;
      }
    fc_store_row<4>(hbuf + row * FC_D, lh, v);
  }
}

// =====================================================================================================================
// forward B: h -> w_2 (+ a1 xhat1 + b1) -> LN3 -> [xhat2, rinv2] -> affine -> linear 128 -> 64 (+ xin) -> [out]
// =====================================================================================================================
__global__ __launch_bounds__(FC_THREADS, 1) void fe_fwd_b_kernel(
    const float* __restrict__ hbuf, const float* __restrict__ xhat1, const float* __restrict__ a1,
    const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ bb2,
    const float* __restrict__ a3, const float* __restrict__ b3, const float* __restrict__ Wl,
    const float* __restrict__ bl, const float* __restrict__ xin, float* __restrict__ xhat2, float* __restrict__ rinv2,
    float* __restrict__ out, int ntiles, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* W2h = reinterpret_cast<__bf16*>(fc_smem);
  __bf16* W2l = W2h + WSZ128;
  __bf16* Wlh = W2l + WSZ128;            // [64][136]
  __bf16* Wll = Wlh + 64 * KP128;
  float* va1 = reinterpret_cast<float*>(Wll + 64 * KP128);
  float* vb1 = va1 + FC_D;
  float* vbb2 = vb1 + FC_D;
  float* va3 = vbb2 + FC_D;
  float* vb3 = va3 + FC_D;
  float* vbl = vb3 + FC_D;               // 64
  {
    FcStageW<FC_D, FC_D> s0;
    FcStageW<FC_D, 64> s1;
    s0.load(W2);
    s1.load(Wl);
    const float x0 = fc_vec_load(a1, FC_D), x1 = fc_vec_load(b1, FC_D), x2 = fc_vec_load(bb2, FC_D), x3 = fc_vec_load(a3, FC_D),
                x4 = fc_vec_load(b3, FC_D), x5 = fc_vec_load(bl, 64);
    s0.store(W2h, W2l);
    s1.store(Wlh, Wll);
    fc_vec_store(va1, FC_D, x0);
    fc_vec_store(vb1, FC_D, x1);
    fc_vec_store(vbb2, FC_D, x2);
    fc_vec_store(va3, FC_D, x3);
    fc_vec_store(vb3, FC_D, x4);
    fc_vec_store(vbl, 64, x5);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 v[4], res[4];
    fc_load_row<4>(hbuf + row * FC_D, lh, v);
    fc_load_row<4>(xhat1 + row * FC_D, lh, res);
    fc_bf16x8 xh[8], xl[8];
    fc_frags<4>(v, xh, xl);
    const int z = fc_opaque_zero();
    fc_affine(res, va1 + z, vb1 + z, lh);                 // r1 = a1 xhat1 + b1: the residual of this sub-layer
    fc_load_vec<4>(vbb2 + z, lh, v);
    fc_gemm<4, 8, KP128>(W2h + z, W2l + z, 0, li, lh, xh, xl, v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[j][r] += res[j][r];
    const float rinv = fc_ln_fwd(v, eps);
    fc_store_row<4>(xhat2 + row * FC_D, lh, v);
    if (lh == 0) rinv2[row] = rinv;
    f32x16 xr[2];
    if (xin) fc_load_row<2>(xin + row * 64, lh, xr);      // the block input (residual of the whole block)
    fc_affine(v, va3 + z, vb3 + z, lh);
    fc_frags<4>(v, xh, xl);
    f32x16 o[2];
    fc_load_vec<2>(vbl + z, lh, o);
    fc_gemm<2, 8, KP128>(Wlh + z, Wll + z, 0, li, lh, xh, xl, o);
    if (xin) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] += xr[j][r];
    }
    fc_store_row<2>(out + row * 64, lh, o);
  }
}

// =====================================================================================================================
// backward A: d_out -> linear^T -> LN3 backward -> [d_s2] -> w_2^T -> relu / dropout mask from h -> [d_hpre]
// =====================================================================================================================
__global__ __launch_bounds__(FC_THREADS, 1) void fe_bwd_a_kernel(
    const float* __restrict__ dout, const float* __restrict__ Wl, const float* __restrict__ xhat2,
    const float* __restrict__ rinv2, const float* __restrict__ a3, const float* __restrict__ W2,
    const float* __restrict__ hbuf, float* __restrict__ ds2, float* __restrict__ dhpre, int ntiles, float eps,
    float keep_scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* Wlth = reinterpret_cast<__bf16*>(fc_smem);      // [128][72]: Wl^T
  __bf16* Wltl = Wlth + FC_D * KP64;
  __bf16* W2th = Wltl + FC_D * KP64;                      // [128][136]: W2^T
  __bf16* W2tl = W2th + WSZ128;
  float* va3 = reinterpret_cast<float*>(W2tl + WSZ128);
  {
    FcStageWt<64, FC_D> s0;
    FcStageWt<FC_D, FC_D> s1;
    s0.load(Wl, FC_D);
    s1.load(W2, FC_D);
    const float x0 = fc_vec_load(a3, FC_D);
    s0.store(Wlth, Wltl);
    s1.store(W2th, W2tl);
    fc_vec_store(va3, FC_D, x0);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 d2[2], xn[4], d[4];
    fc_load_row<2>(dout + row * 64, lh, d2);
    fc_load_row<4>(xhat2 + row * FC_D, lh, xn);
    const float rinv = rinv2[row];
    fc_bf16x8 yh[4], yl[4];
    fc_frags<2>(d2, yh, yl);
    const int z = fc_opaque_zero();
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[j][r] = 0.f;
    fc_gemm<4, 4, KP64>(Wlth + z, Wltl + z, 0, li, lh, yh, yl, d);
    fc_ln_bwd(d, xn, va3 + z, lh, rinv, eps);
    fc_store_row<4>(ds2 + row * FC_D, lh, d);
    fc_bf16x8 xh[8], xl[8];
    fc_frags<4>(d, xh, xl);
    fc_load_row<4>(hbuf + row * FC_D, lh, xn);             // h: zero where relu or dropout zeroed the output
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[j][r] = 0.f;
    fc_gemm<4, 8, KP128>(W2th + z, W2tl + z, 0, li, lh, xh, xl, d);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[j][r] = xn[j][r] > 0.f ? d[j][r] * keep_scale : 0.f;
    fc_store_row<4>(dhpre + row * FC_D, lh, d);
  }
}

// =====================================================================================================================
// backward B: d_hpre -> w_1^T (+ d_s2) -> LN1 backward -> [d_s1] -> O-proj^T -> [d_ctx]
// =====================================================================================================================
__global__ __launch_bounds__(FC_THREADS, 1) void fe_bwd_b_kernel(
    const float* __restrict__ dhpre, const float* __restrict__ ds2, const float* __restrict__ W1,
    const float* __restrict__ xhat1, const float* __restrict__ rinv1, const float* __restrict__ a1,
    const float* __restrict__ Wo, float* __restrict__ ds1, float* __restrict__ dctx, int ntiles, float eps,
    const float* __restrict__ octx, float* __restrict__ Dw, int ntok, __bf16* __restrict__ dop, long pls, float gmul) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* W1th = reinterpret_cast<__bf16*>(fc_smem);
  __bf16* W1tl = W1th + WSZ128;
  __bf16* Woth = W1tl + WSZ128;
  __bf16* Wotl = Woth + WSZ128;
  float* va1 = reinterpret_cast<float*>(Wotl + WSZ128);
  {
    FcStageWt<FC_D, FC_D> s0, s1;
    s0.load(W1, FC_D);
    s1.load(Wo, FC_D);
    const float x0 = fc_vec_load(a1, FC_D);
    s0.store(W1th, W1tl);
    s1.store(Woth, Wotl);
    fc_vec_store(va1, FC_D, x0);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 d[4], xn[4];
    fc_load_row<4>(dhpre + row * FC_D, lh, d);
    fc_load_row<4>(xhat1 + row * FC_D, lh, xn);
    const float rinv = rinv1[row];
    fc_bf16x8 xh[8], xl[8];
    fc_frags<4>(d, xh, xl);
    fc_load_row<4>(ds2 + row * FC_D, lh, d);               // gradient of r1 through the LN3 residual slot
    const int z = fc_opaque_zero();
    fc_gemm<4, 8, KP128>(W1th + z, W1tl + z, 0, li, lh, xh, xl, d);
    fc_ln_bwd(d, xn, va1 + z, lh, rinv, eps);
    fc_store_row<4>(ds1 + row * FC_D, lh, d);
    fc_frags<4>(d, xh, xl);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) d[j][r] = 0.f;
    fc_gemm<4, 8, KP128>(Woth + z, Wotl + z, 0, li, lh, xh, xl, d);
    if (dctx) fc_store_row<4>(dctx + row * FC_D, lh, d);
    if (dop) fc_store_planes<4>(dop + row * 256, lh, d, gmul);           // dO / P(keep), pre-split for the attention backward
    if (Dw) {
      // D[b][head][token] = sum over the head's 32 columns of dO * O (the attention backward's row term): accumulator
      // tile j IS head j, so this is an in-lane sum + one exchange -- the separate prep pass over dO and O is gone
      fc_load_row<4>(octx + row * FC_D, lh, xn);
      const size_t bi = row / (size_t)ntok, n = row - bi * (size_t)ntok;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float sdot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sdot += d[j][r] * xn[j][r];
        sdot = fc_rowsum(sdot);
        if (lh == 0) Dw[(bi * 4 + j) * (size_t)ntok + n] = sdot;
      }
    }
  }
}

// =====================================================================================================================
// forward QKV: tok = [feat | pe[row % ntok]] -> [tok] (column group 0 only) ; qkv[:, 128 y .. 128 y + 127] = tok Wqkv_y^T + b
// (grid.y = 3 column groups of the packed projection; the concat pass and its 67 MB round trip are gone: the token is
// assembled in registers from the 64 feature columns and the positional-encoding table, tbsrn.py:83-86)
// =====================================================================================================================
__global__ __launch_bounds__(256, 2) void fe_qkv_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ pe,
                                                           const float* __restrict__ Wqkv,
                                                           const float* __restrict__ bqkv, float* __restrict__ tok,
                                                           float* __restrict__ qkv, int ntiles, int ntok,
                                                           __bf16* __restrict__ planes, long pls, float qmul,
                                                           const float* __restrict__ bn_gamma,
                                                           const float* __restrict__ bn_beta,
                                                           const float* __restrict__ bn_mean,
                                                           const float* __restrict__ bn_invstd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* Wh = reinterpret_cast<__bf16*>(fc_smem);
  __bf16* Wl = Wh + WSZ128;
  float* vb = reinterpret_cast<float*>(Wl + WSZ128);
  float* vbn = vb + FC_D;                                  // bn_gamma != nullptr: gamma | mean | invstd | beta, 64 each
  const int y = blockIdx.y;
  {  // two-phase staging with this kernel's 256 threads: the weight block's 16 float4 per thread, the bias and the four
     // BatchNorm vectors are requested together
    FcStageW<FC_D, FC_D, 256> s0;
    s0.load(Wqkv + (size_t)y * FC_D * FC_D);
    const int i4 = min((int)threadIdx.x, FC_D - 1), i6 = min((int)threadIdx.x, 63);
    const float xb = bqkv ? bqkv[y * FC_D + i4] : 0.f;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (bn_gamma) { g0 = bn_gamma[i6]; g1 = bn_mean[i6]; g2 = bn_invstd[i6]; g3 = bn_beta[i6]; }
    s0.store(Wh, Wl);
    if (threadIdx.x < FC_D) vb[threadIdx.x] = xb;
    if (bn_gamma && threadIdx.x < 64) {
      vbn[threadIdx.x] = g0;
      vbn[64 + threadIdx.x] = g1;
      vbn[128 + threadIdx.x] = g2;
      vbn[192 + threadIdx.x] = g3;
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 4 + wave; t < ntiles; t += gridDim.x * 4) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 v[4];
    {
      f32x16 a[2], b[2];
      fc_load_row<2>(feat + row * 64, lh, a);
      fc_load_row<2>(pe + (row % (size_t)ntok) * 64, lh, b);
      if (bn_gamma) {
        // `feat` is the block's second convolution BEFORE its BatchNorm (tbsrn.py:246-251): normalised here, on load, with
        // bn_apply_kernel's expression -- the 33.5 MB normalised tensor and its launch are gone (its only consumer was this load)
        const int z0 = fc_opaque_zero();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 g_[1], mu[1], is[1], be[1];
          fc_load_vec<1>(vbn + 32 * j + z0, lh, g_);
          fc_load_vec<1>(vbn + 64 + 32 * j + z0, lh, mu);
          fc_load_vec<1>(vbn + 128 + 32 * j + z0, lh, is);
          fc_load_vec<1>(vbn + 192 + 32 * j + z0, lh, be);
#pragma unroll
          for (int r = 0; r < 16; ++r) a[j][r] = g_[0][r] * (a[j][r] - mu[0][r]) * is[0][r] + be[0][r];
        }
      }
      v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
    }
    if (y == 0) fc_store_row<4>(tok + row * FC_D, lh, v);
    fc_bf16x8 xh[8], xl[8];
    fc_frags<4>(v, xh, xl);
    const int z = fc_opaque_zero();
    fc_load_vec<4>(vb + z, lh, v);
    fc_gemm<4, 8, KP128>(Wh + z, Wl + z, 0, li, lh, xh, xl, v);
    if (qkv) fc_store_row<4>(qkv + row * 384 + 128 * y, lh, v);
    // pre-split bf16 hi / lo planes of Q (x scale log2 e), K, V for the attention kernels' PL variants
    if (planes) fc_store_planes<4>(planes + row * 768 + 256 * y, lh, v, y == 0 ? qmul : 1.f);   // packed [rows][Q | K | V]
  }
}

// =====================================================================================================================
// backward QKV: d_feat[rows, 64] = dqkv[rows, 384] . Wqkv[384, 0:64] + d_s1[rows, 0:64]
// (the token is [feature | positional encoding]; the encoding half is a constant: its 64 gradient columns have no
// consumer, so half of the former K = 384 -> 128 data-gradient GEMM and its 67 MB output are gone)
// =====================================================================================================================
__global__ __launch_bounds__(FC_THREADS, 1) void fe_bwd_qkv_kernel(const float* __restrict__ dqkv,
                                                                   const float* __restrict__ Wqkv,
                                                                   const float* __restrict__ ds1,
                                                                   float* __restrict__ dfeat, int ntiles, int ld_ds1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fc_smem[];
  __bf16* Wth = reinterpret_cast<__bf16*>(fc_smem);        // [64][392]
  __bf16* Wtl = Wth + 64 * KP384;
  {
    FcStageWt<384, 64> s0;
    s0.load(Wqkv, FC_D);
    s0.store(Wth, Wtl);
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  for (int t = blockIdx.x * 8 + wave; t < ntiles; t += gridDim.x * 8) {
    const size_t row = (size_t)t * 32 + li;
    f32x16 o[2];
    if (ds1) {
      fc_load_row<2>(ds1 + row * ld_ds1, lh, o);
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    }
    const int z = fc_opaque_zero();
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      f32x16 v[4];
      fc_load_row<4>(dqkv + row * 384 + 128 * q, lh, v);
      fc_bf16x8 xh[8], xl[8];
      fc_frags<4>(v, xh, xl);
      fc_gemm<2, 8, KP384>(Wth + z, Wtl + z, 128 * q, li, lh, xh, xl, o);
    }
    fc_store_row<2>(dfeat + row * 64, lh, o);
  }
}

// =====================================================================================================================
// Parameter gradients of  LayerNorm -> Linear  from ONE weight-gradient GEMM on the normalised rows.
//   r = a xhat + b,  y = W r + c  (W [N][128]).  With G[o][k] = sum_m dy[m][o] xhat[m][k] and dc[o] = sum_m dy[m][o]:
//     dW[o][k] = sum_m dy[m][o] r[m][k]                 = a[k] G[o][k] + b[k] dc[o]
//     da[k]    = sum_m (dy W)[m][k] xhat[m][k]          = sum_o W[o][k] G[o][k]   (+ extra[k]:       other consumers of r)
//     db[k]    = sum_m (dy W)[m][k]                     = sum_o W[o][k] dc[o]     (+ extra[128 + k])
// G | dc: the output of the ordinary weight-gradient kernel run with xhat as its X operand.
// =====================================================================================================================
__global__ __launch_bounds__(1024) void fe_ln_lin_finish_kernel(const float* __restrict__ G, const float* __restrict__ W,
                                                               const float* __restrict__ a, const float* __restrict__ b,
                                                               const float* __restrict__ extra, float* __restrict__ dW,
                                                               float* __restrict__ dc, float* __restrict__ da,
                                                               float* __restrict__ db, int N) {
  // grid: 4 blocks, each 32 columns k x 32 groups of rows o (fixed-order fold of the groups through LDS)
  __shared__ float red[2][32][33];
  const int kl = threadIdx.x & 31, og = threadIdx.x >> 5;
  const int k = blockIdx.x * 32 + kl;
  const float* Gc = G + (size_t)N * FC_D;
  const float ak = a[k], bk = b[k];
  float sa = 0.f, sb = 0.f;
  for (int o = og; o < N; o += 32) {
    const float g = G[(size_t)o * FC_D + k], w = W[(size_t)o * FC_D + k], c = Gc[o];
    dW[(size_t)o * FC_D + k] = ak * g + bk * c;
    sa += w * g;
    sb += w * c;
  }
  red[0][og][kl] = sa;
  red[1][og][kl] = sb;
  __syncthreads();
  if (og == 0) {
    float ta = extra ? extra[k] : 0.f, tb = extra ? extra[FC_D + k] : 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      ta += red[0][q][kl];
      tb += red[1][q][kl];
    }
    da[k] = ta;
    db[k] = tb;
    if (dc && k < N) dc[k] = Gc[k];
  }
}

// column sums of A .* B and of A over the rows ([rows][128] each): per-block partials, folded in block order
#define FC_CS_BLOCKS 256
__global__ __launch_bounds__(256) void fe_colsum2_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ part, long rows) {
  __shared__ float4 red[2][8][32];
  const int q = threadIdx.x & 31, rg = threadIdx.x >> 5;
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
  for (long r = (long)blockIdx.x * 8 + rg; r < rows; r += (long)gridDim.x * 8) {
    const float4 x = reinterpret_cast<const float4*>(A + r * FC_D)[q];
    const float4 y = reinterpret_cast<const float4*>(B + r * FC_D)[q];
    pa.x += x.x * y.x; pa.y += x.y * y.y; pa.z += x.z * y.z; pa.w += x.w * y.w;
    pb.x += x.x; pb.y += x.y; pb.z += x.z; pb.w += x.w;
  }
  red[0][rg][q] = pa;
  red[1][rg][q] = pb;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int w = threadIdx.x >> 5;
    float4 s = red[w][0][q];
#pragma unroll
    for (int g = 1; g < 8; ++g) {
      s.x += red[w][g][q].x; s.y += red[w][g][q].y; s.z += red[w][g][q].z; s.w += red[w][g][q].w;
    }
    reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 256 + w * FC_D)[q] = s;
  }
}
__global__ __launch_bounds__(256) void fe_colsum2_fold_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nblocks) {
  // grid: 8 blocks x 32 outputs; 8 thread groups each sum a contiguous range of the partials, folded in group order
  __shared__ float red[8][33];
  const int il = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + il;
  const int per = (nblocks + 7) / 8, b0 = g * per, b1 = min(nblocks, b0 + per);
  float s = 0.f;
#pragma unroll 8
  for (int b = b0; b < b1; ++b) s += part[(size_t)b * 256 + i];
  red[g][il] = s;
  __syncthreads();
  if (g == 0) {
    float t = red[0][il];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += red[q][il];
    out[i] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int focr_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int Cin,
                                 int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx, int prezeroed, float* ws,
                                 long ws_floats, hipStream_t stream);
extern "C" long focr_conv2d_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW);

namespace {
constexpr size_t FC_LDS_FWD_A = (size_t)4 * WSZ128 * 2 + 4 * FC_D * 4;
constexpr size_t FC_LDS_FWD_B = (size_t)(2 * WSZ128 + 2 * 64 * KP128) * 2 + (5 * FC_D + 64) * 4;
constexpr size_t FC_LDS_BWD_A = (size_t)(2 * FC_D * KP64 + 2 * WSZ128) * 2 + FC_D * 4;
constexpr size_t FC_LDS_BWD_B = (size_t)4 * WSZ128 * 2 + FC_D * 4;
constexpr size_t FC_LDS_BWD_QKV = (size_t)2 * 64 * KP384 * 2;
constexpr size_t FC_LDS_QKV_FWD = (size_t)2 * WSZ128 * 2 + FC_D * 4 + 4 * 64 * 4;

int fc_blocks(int ntiles) {
  int nb = (ntiles + 7) / 8;
  return nb > 256 ? 256 : nb;
}
template <class Kern>
int fc_set_lds(Kern kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess;
}
}  // namespace

// FeatureEnhancer eligibility of the fused chains: bf16x3 arithmetic only (precision 0 keeps the per-op fp32 path)
extern "C" int focr_fe_chain_supported(long rows, int d_model) {
  return focr_get_precision() != 0 && d_model == FC_D && rows > 0 && rows % 32 == 0 && rows < (1l << 31) - 64;
}

extern "C" int focr_fe_post_fwd(const float* ctx, const float* tok, const float* xin, const float* wo, const float* bo,
                                const float* a1, const float* b1, const float* w1, const float* bb1, const float* w2,
                                const float* bb2, const float* a3, const float* b3, const float* wl, const float* bl,
                                float* xhat1, float* rinv1, float* h, float* xhat2, float* rinv2, float* out, long rows,
                                float eps, float p_drop, uint64_t seed, float* keep_scale, hipStream_t stream) {
  FOCR_CHECK_ARG(ctx && tok && wo && a1 && b1 && w1 && w2 && a3 && b3 && wl && xhat1 && rinv1 && h && xhat2 && rinv2 &&
                     out && keep_scale,
                 "null pointer");
  FOCR_CHECK_ARG(focr_fe_chain_supported(rows, FC_D), "rows must be a positive multiple of 32 (precision mode != 0)");
  FOCR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  uint32_t kq = 0u;
  *keep_scale = 1.f;
  if (p_drop > 0.f) {
    kq = 65536u - (uint32_t)(p_drop * 65536.0f + 0.5f);
    FOCR_CHECK_ARG(kq > 0u, "dropout probability rounds to 1");
    *keep_scale = 65536.f / (float)kq;
    if (kq >= 65536u) kq = 0u;
  }
  static focr_dev_flags attr;
  if (focr_dev_first(attr)) {
    if (!fc_set_lds(fe_fwd_a_kernel, FC_LDS_FWD_A) || !fc_set_lds(fe_fwd_b_kernel, FC_LDS_FWD_B)) {
      focr_set_error("focr_fe_post_fwd: cannot reserve %zu bytes of LDS", FC_LDS_FWD_A);
      return FOCR_EHIP;
    }
    focr_dev_mark(attr);
  }
  const int ntiles = (int)(rows / 32), nb = fc_blocks(ntiles);
  hipLaunchKernelGGL(fe_fwd_a_kernel, dim3(nb), FC_THREADS, FC_LDS_FWD_A, stream, ctx, tok, wo, bo, a1, b1, w1, bb1,
                     xhat1, rinv1, h, ntiles, eps, kq, *keep_scale, (uint32_t)(seed ^ (seed >> 32)), focr_seed_epoch());
  hipLaunchKernelGGL(fe_fwd_b_kernel, dim3(nb), FC_THREADS, FC_LDS_FWD_B, stream, (const float*)h, (const float*)xhat1,
                     a1, b1, w2, bb2, a3, b3, wl, bl, xin, xhat2, rinv2, out, ntiles, eps);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_fe_post_bwd(const float* d_out, const float* wl, const float* xhat2, const float* rinv2,
                                const float* a3, const float* w2, const float* h, float keep_scale, const float* w1,
                                const float* xhat1, const float* rinv1, const float* a1, const float* wo, float* d_s2,
                                float* d_hpre, float* d_s1, float* d_ctx, long rows, float eps, const float* ctx,
                                float* dwork, int ntok, void* d_ctx_planes, float planes_mul, hipStream_t stream) {
  FOCR_CHECK_ARG(d_out && wl && xhat2 && rinv2 && a3 && w2 && h && w1 && xhat1 && rinv1 && a1 && wo && d_s2 && d_hpre &&
                     d_s1 && (d_ctx || d_ctx_planes),
                 "null pointer");
  FOCR_CHECK_ARG(focr_fe_chain_supported(rows, FC_D), "rows must be a positive multiple of 32 (precision mode != 0)");
  FOCR_CHECK_ARG(!dwork || (ctx && ntok > 0 && rows % ntok == 0), "dwork needs ctx and the tokens per image");
  static focr_dev_flags attr;
  if (focr_dev_first(attr)) {
    if (!fc_set_lds(fe_bwd_a_kernel, FC_LDS_BWD_A) || !fc_set_lds(fe_bwd_b_kernel, FC_LDS_BWD_B)) {
      focr_set_error("focr_fe_post_bwd: cannot reserve %zu bytes of LDS", FC_LDS_BWD_B);
      return FOCR_EHIP;
    }
    focr_dev_mark(attr);
  }
  const int ntiles = (int)(rows / 32), nb = fc_blocks(ntiles);
  hipLaunchKernelGGL(fe_bwd_a_kernel, dim3(nb), FC_THREADS, FC_LDS_BWD_A, stream, d_out, wl, xhat2, rinv2, a3, w2, h,
                     d_s2, d_hpre, ntiles, eps, keep_scale);
  hipLaunchKernelGGL(fe_bwd_b_kernel, dim3(nb), FC_THREADS, FC_LDS_BWD_B, stream, (const float*)d_hpre,
                     (const float*)d_s2, w1, xhat1, rinv1, a1, wo, d_s1, d_ctx, ntiles, eps, ctx, dwork, ntok,
                     reinterpret_cast<__bf16*>(d_ctx_planes), 0L, planes_mul);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// tok [rows,128] = [feat | pe[row % ntok]], qkv [rows,384] = tok Wqkv^T + bqkv (packed q | k | v projection)
// focr_fe_qkv_fwd_bn: `feat` is the input of a train-mode BatchNorm2d(64) whose statistics are final (save_mean / save_invstd
// of focr_bn_train_fwd_stats with y = NULL): the normalisation gamma (x - mean) invstd + beta is applied on load
extern "C" int focr_fe_qkv_fwd_bn(const float* feat, const float* pe, const float* wqkv, const float* bqkv, float* tok,
                                  float* qkv, long rows, int ntok, void* planes, float q_mul, const float* bn_gamma,
                                  const float* bn_beta, const float* bn_mean, const float* bn_invstd, hipStream_t stream);
extern "C" int focr_fe_qkv_fwd(const float* feat, const float* pe, const float* wqkv, const float* bqkv, float* tok,
                               float* qkv, long rows, int ntok, void* planes, float q_mul, hipStream_t stream) {
  return focr_fe_qkv_fwd_bn(feat, pe, wqkv, bqkv, tok, qkv, rows, ntok, planes, q_mul, nullptr, nullptr, nullptr, nullptr, stream);
}
extern "C" int focr_fe_qkv_fwd_bn(const float* feat, const float* pe, const float* wqkv, const float* bqkv, float* tok,
                                  float* qkv, long rows, int ntok, void* planes, float q_mul, const float* bn_gamma,
                                  const float* bn_beta, const float* bn_mean, const float* bn_invstd, hipStream_t stream) {
  FOCR_CHECK_ARG(feat && pe && wqkv && tok && (qkv || planes) && ntok > 0, "bad argument");
  FOCR_CHECK_ARG(!bn_gamma || (bn_beta && bn_mean && bn_invstd), "BatchNorm vectors: all four or none");
  FOCR_CHECK_ARG(focr_fe_chain_supported(rows, FC_D), "rows must be a positive multiple of 32 (precision mode != 0)");
  static focr_dev_flags attr;
  if (focr_dev_first(attr)) {
    if (!fc_set_lds(fe_qkv_fwd_kernel, FC_LDS_QKV_FWD)) {
      focr_set_error("focr_fe_qkv_fwd: cannot reserve %zu bytes of LDS", FC_LDS_QKV_FWD);
      return FOCR_EHIP;
    }
    focr_dev_mark(attr);
  }
  const int ntiles = (int)(rows / 32);
  int nb = (ntiles + 3) / 4;
  if (nb > 171) nb = 171;                   // x 3 column groups = 513 blocks: one round at two blocks per CU
  hipLaunchKernelGGL(fe_qkv_fwd_kernel, dim3(nb, 3), 256, FC_LDS_QKV_FWD, stream, feat, pe, wqkv, bqkv, tok, qkv, ntiles,
                     ntok, reinterpret_cast<__bf16*>(planes), rows * 256, q_mul, bn_gamma, bn_beta, bn_mean, bn_invstd);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_fe_qkv_dgrad(const float* dqkv, const float* wqkv, const float* d_s1, float* d_feat, long rows,
                                 hipStream_t stream) {
  FOCR_CHECK_ARG(dqkv && wqkv && d_feat, "null pointer");
  FOCR_CHECK_ARG(focr_fe_chain_supported(rows, FC_D), "rows must be a positive multiple of 32 (precision mode != 0)");
  static focr_dev_flags attr;
  if (focr_dev_first(attr)) {
    if (!fc_set_lds(fe_bwd_qkv_kernel, FC_LDS_BWD_QKV)) {
      focr_set_error("focr_fe_qkv_dgrad: cannot reserve %zu bytes of LDS", FC_LDS_BWD_QKV);
      return FOCR_EHIP;
    }
    focr_dev_mark(attr);
  }
  const int ntiles = (int)(rows / 32), nb = fc_blocks(ntiles);
  hipLaunchKernelGGL(fe_bwd_qkv_kernel, dim3(nb), FC_THREADS, FC_LDS_BWD_QKV, stream, dqkv, wqkv, d_s1, d_feat, ntiles,
                     FC_D);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// All parameter gradients of the block's row-local layers in one call (the caller runs it on its weight-gradient side
// stream).  Gradient targets are OVERWRITTEN (every parameter is used once per step).
//   ws: scratch, focr_fe_wgrads_ws_floats(rows) floats.
extern "C" long focr_fe_wgrads_ws_floats(long rows) {
  long a = focr_conv2d_wgrad_ws_floats((int)rows, 1, 1, FC_D, 384, 1, 1, 0, 0);
  long b = focr_conv2d_wgrad_ws_floats((int)rows, 1, 1, FC_D, FC_D, 1, 1, 0, 0);
  long c = focr_conv2d_wgrad_ws_floats((int)rows, 1, 1, FC_D, 64, 1, 1, 0, 0);
  long m = a > b ? a : b;
  if (c > m) m = c;
  // + G | dc of one LayerNorm-fed linear (128 x 128 + 128), the column-sum partials and their fold
  return m + (FC_D * FC_D + FC_D) + (long)FC_CS_BLOCKS * 256 + 256 + 64;
}

extern "C" int focr_fe_wgrads(const float* d_out, const float* xhat2, const float* d_s2, const float* h,
                              const float* d_hpre, const float* xhat1, const float* d_s1, const float* ctx,
                              const float* dqkv, const float* tok, const float* wl, const float* w1, const float* a1,
                              const float* b1, const float* a3, const float* b3, float* g_wl, float* g_bl, float* g_a3,
                              float* g_b3, float* g_w2, float* g_bb2, float* g_w1, float* g_bb1, float* g_a1,
                              float* g_b1, float* g_wo, float* g_bo, float* g_wqkv, float* g_bqkv, float* ws,
                              long ws_floats, long rows, int parts, hipStream_t stream) {
  // parts: bit 0 = the four linears that only need the backward chains' outputs (O-proj, w_2, LN1 -> w_1, LN3 -> linear);
  // bit 1 = the packed q | k | v projection, which needs the attention backward's dqkv.  The caller issues part 1 BEFORE
  // the attention backward (VALU-bound, light on HBM: a good partner for these streaming kernels) and part 2 after it.
  FOCR_CHECK_ARG(parts >= 1 && parts <= 3 && ws, "parts must be 1, 2 or 3");
  FOCR_CHECK_ARG(!(parts & 1) || (d_out && xhat2 && d_s2 && h && d_hpre && xhat1 && d_s1 && ctx && wl && w1 && a1 && b1 &&
                                  a3 && b3 && g_wl && g_bl && g_a3 && g_b3 && g_w2 && g_bb2 && g_w1 && g_bb1 && g_a1 &&
                                  g_b1 && g_wo && g_bo),
                 "null pointer (part 1)");
  FOCR_CHECK_ARG(!(parts & 2) || (dqkv && tok && g_wqkv && g_bqkv), "null pointer (part 2)");
  FOCR_CHECK_ARG(focr_fe_chain_supported(rows, FC_D), "rows must be a positive multiple of 32 (precision mode != 0)");
  FOCR_CHECK_ARG(ws_floats >= focr_fe_wgrads_ws_floats(rows), "workspace too small");
  const long tail = (FC_D * FC_D + FC_D) + (long)FC_CS_BLOCKS * 256 + 256 + 64;
  const long wsg = ws_floats - tail;                   // scratch of the weight-gradient kernels
  float* G = ws + wsg;                                 // [N][128] | dc[N]
  float* part = G + (FC_D * FC_D + FC_D);
  float* extra = part + (long)FC_CS_BLOCKS * 256;
  const int M = (int)rows;
  int rc;
  // packed q/k/v projection: X = tok, dY = dqkv
  if (parts & 2)
    if ((rc = focr_conv2d_wgrad(tok, dqkv, g_wqkv, g_bqkv, M, 1, 1, FC_D, 384, 1, 1, 0, 0, 0, 0, 0, ws, wsg, stream))) return rc;
  if (!(parts & 1)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  // O-proj: X = ctx, dY = d_s1 (the gradient of LN1's input sum)
  if ((rc = focr_conv2d_wgrad(ctx, d_s1, g_wo, g_bo, M, 1, 1, FC_D, FC_D, 1, 1, 0, 0, 0, 0, 0, ws, wsg, stream))) return rc;
  // w_2: X = h, dY = d_s2
  if ((rc = focr_conv2d_wgrad(h, d_s2, g_w2, g_bb2, M, 1, 1, FC_D, FC_D, 1, 1, 0, 0, 0, 0, 0, ws, wsg, stream))) return rc;
  // LN1 -> w_1: G = d_hpre^T xhat1; r1 also feeds the LN3 residual slot: extra = colsum(d_s2 .* xhat1), colsum(d_s2)
  if ((rc = focr_conv2d_wgrad(xhat1, d_hpre, G, G + FC_D * FC_D, M, 1, 1, FC_D, FC_D, 1, 1, 0, 0, 0, 0, 0, ws, wsg, stream)))
    return rc;
  hipLaunchKernelGGL(fe_colsum2_kernel, dim3(FC_CS_BLOCKS), 256, 0, stream, d_s2, xhat1, part, rows);
  hipLaunchKernelGGL(fe_colsum2_fold_kernel, dim3(8), 256, 0, stream, (const float*)part, extra, FC_CS_BLOCKS);
  hipLaunchKernelGGL(fe_ln_lin_finish_kernel, dim3(4), 1024, 0, stream, (const float*)G, w1, a1, b1, (const float*)extra,
                     g_w1, g_bb1, g_a1, g_b1, FC_D);
  // LN3 -> linear 128 -> 64: G = d_out^T xhat2
  if ((rc = focr_conv2d_wgrad(xhat2, d_out, G, G + 64 * FC_D, M, 1, 1, FC_D, 64, 1, 1, 0, 0, 0, 0, 0, ws, wsg, stream)))
    return rc;
  hipLaunchKernelGGL(fe_ln_lin_finish_kernel, dim3(4), 1024, 0, stream, (const float*)G, wl, a3, b3, (const float*)nullptr,
                     g_wl, g_bl, g_a3, g_b3, 64);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

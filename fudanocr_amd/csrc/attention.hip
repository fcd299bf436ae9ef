// Fused multi-head self-attention for the TBSRN FeatureEnhancer (head dim 32), fp32 MFMA.
//
// Replaces reference model/tbsrn.py:132-150 (`attention`: matmul(Q,K^T)/sqrt(d_k) -> softmax
// -> Dropout(0.1) -> matmul(P,V)), which materialises a [B,4,1024,1024] score tensor
// (16.8 MB per image per SRB).  Here the scores never leave registers (online softmax).
//
// Operand orientation ("swapped" products) is chosen so that every per-query quantity
// (running max, running sum, LSE, D = rowsum(dO*O)) is lane-local:
//   S^T = K Q^T   : MFMA A = K rows (from LDS), B = Q (registers)  -> lane = query, regs = keys
//   O^T = V^T P^T : MFMA A = V[key][d] (LDS, one float per lane), B = P (the S^T accumulator
//                   itself: reg s of lane (q,half) IS the B operand of MFMA step s)
// so P is never shuffled or written anywhere.  The K index of the f32 32x32x2 MFMA is
// free to permute, the only rule is that A and B use the same permutation.
//
// Layout: q,k,v,o,do,dq,dk,dv are [B, Ntok, ld] fp32 with head h at columns h*32..h*32+31.
// LSE, D are [B, H, Ntok].  Dropout keep-bits: uint32 [B, H, Ntok, Ntok/32] (bit = key % 32), written by
// the forward kernel (two 16-bit draws per counter hash), read by both backward kernels.
#include "focr_common.h"

#define KP 36   // LDS pitch (floats) of tiles read with ds_read_b128 fragments

__device__ __forceinline__ int key_of(int s, int lh) { return (s & 3) + 8 * (s >> 2) + 4 * lh; }

template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ Q,
                                                       const float* __restrict__ K,
                                                       const float* __restrict__ V,
                                                       float* __restrict__ O, float* __restrict__ LSE,
                                                       const uint32_t* __restrict__ MASK, int Ntok, int ld,
                                                       float scale, float p_drop, uint64_t seed, int nheads) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * KP];
  __shared__ __attribute__((aligned(16))) float Vs[64 * 32];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const int q = qb_ * 128 + wave * 32 + li;

  float4 qf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float4 v = *reinterpret_cast<const float4*>(Q + base + (size_t)q * ld + 8 * t + 4 * lh);
    qf[t] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
  }
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m = -1e30f, l = 0.f;
  // dropout: one 32-bit hash per PAIR of adjacent keys, 16 bits each (p quantised to 1/65536);
  // the keep bits are packed (bit = key % 32) and written out for the two backward kernels
  const uint32_t thr = DROPOUT ? (uint32_t)(p_drop * 65536.0f + 0.5f) : 0u;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)thr / 65536.f) : 1.f;
  const uint32_t* mrow = MASK + ((size_t)(b * H + h) * Ntok + q) * (size_t)(Ntok / 32);

  // staging: 64 rows x 8 float4 per tile, thread -> (row = idx>>3, c4 = idx&7), idx = tid+256*i
  // staging registers (named scalars: arrays captured by a lambda end up in scratch)
  float4 kreg0, kreg1, vreg0, vreg1;
  const int srow = tid >> 3, scol = (tid & 7) * 4;      // rows srow and srow+32
#define LOAD_TILE(kt)                                                                   \
  do {                                                                                  \
    size_t off0_ = base + (size_t)((kt) * 64 + srow) * ld + scol;                       \
    size_t off1_ = off0_ + (size_t)32 * ld;                                             \
    kreg0 = *reinterpret_cast<const float4*>(K + off0_);                                \
    kreg1 = *reinterpret_cast<const float4*>(K + off1_);                                \
    vreg0 = *reinterpret_cast<const float4*>(V + off0_);                                \
    vreg1 = *reinterpret_cast<const float4*>(V + off1_);                                \
  } while (0)
  const int ntiles = Ntok / 64;
  LOAD_TILE(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    *reinterpret_cast<float4*>(&Ks[srow * KP + scol]) = kreg0;
    *reinterpret_cast<float4*>(&Ks[(srow + 32) * KP + scol]) = kreg1;
    *reinterpret_cast<float4*>(&Vs[srow * 32 + scol]) = vreg0;
    *reinterpret_cast<float4*>(&Vs[(srow + 32) * 32 + scol]) = vreg1;
    __syncthreads();
    if (kt + 1 < ntiles) LOAD_TILE(kt + 1);
    uint2 mw = make_uint2(0u, 0u);
    if (DROPOUT) mw = *reinterpret_cast<const uint2*>(mrow + kt * 2);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 kf = *reinterpret_cast<const float4*>(&Ks[(sub * 32 + li) * KP + 8 * t + 4 * lh]);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[t].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[t].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[t].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[t].w, s, 0, 0, 0);
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float mn = fmaxf(m, mx);
      float alpha = __expf(m - mn);
      m = mn;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __expf(s[r] - mn);
        ls += p;
        s[r] = p;
      }
      if (DROPOUT) {
        const uint32_t w = sub ? mw.y : mw.x;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = ((w >> key_of(r, lh)) & 1u) ? s[r] * inv_keep : 0.f;
      }
      l = l * alpha + ls;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = Vs[(sub * 32 + key_of(r, lh)) * 32 + li];
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, s[r], oacc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  float inv = 1.f / l;
  float* orow = O + base + (size_t)q * ld;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float4 v = make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv,
                           oacc[4 * g + 3] * inv);
    *reinterpret_cast<float4*>(orow + 8 * g + 4 * lh) = v;
  }
  if (lh == 0) LSE[(size_t)(b * H + h) * Ntok + q] = m + __logf(l);
}

// Dropout keep bits, 1 bit per score: word (row, j) covers keys 32j..32j+31 of query row `row`
// (row = (b*H+h)*Ntok + q).  One counter hash yields two 16-bit draws (adjacent keys); p is quantised to
// 1/65536.  A pure-VALU pre-pass: the three attention kernels then only test bits.
__global__ __launch_bounds__(256) void attn_mask_kernel(uint32_t* __restrict__ mask, long nwords, int wpr,
                                                        float p_drop, uint64_t seed) {
  const uint32_t thr = (uint32_t)(p_drop * 65536.0f + 0.5f);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x) {
    uint32_t row = (uint32_t)(i / wpr), j = (uint32_t)(i % wpr);
    uint32_t rk = rng_rowkey(seed, row);
    uint32_t w = 0u;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      uint32_t hsh = rng_elem(rk, j * 16 + t);
      w |= (((hsh & 0xffffu) >= thr) ? 1u : 0u) << (2 * t);
      w |= (((hsh >> 16) >= thr) ? 1u : 0u) << (2 * t + 1);
    }
    mask[i] = w;
  }
}

// D[b,h,q] = sum_d dO[q][d] * O[q][d]
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ O,
                                                            const float* __restrict__ dO,
                                                            float* __restrict__ D, int Ntok, int ld,
                                                            int H, long total) {
  // one thread per (b, q, h): reads 32 contiguous floats of both
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int h = (int)(i % H);
  long bq = i / H;
  int q = (int)(bq % Ntok);
  int b = (int)(bq / Ntok);
  const float4* o4 = reinterpret_cast<const float4*>(O + (size_t)bq * ld + h * 32);
  const float4* d4 = reinterpret_cast<const float4*>(dO + (size_t)bq * ld + h * 32);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float4 a = o4[j], c = d4[j];
    s += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
  D[(size_t)(b * H + h) * Ntok + q] = s;
}

// pass 1: dK, dV.  block = 128 keys (wave = 32 keys), loop over 64-query tiles.
//   S  [q][key] : A = Qs rows (LDS b128), B = K (regs)      lane = key, regs = queries
//   dP [q][key] : A = dO rows (LDS b128), B = V (regs)
//   dV^T[d][key] = sum_q dO[q][d] Pd[q][key] : A = dO[q(s)][d=li] (LDS b32), B = Pd reg s
//   dK^T[d][key] = sum_q Qs[q][d] dS[q][key] : A = Qs[q(s)][d=li] (LDS b32), B = dS reg s
template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    float* __restrict__ dK, float* __restrict__ dV, const uint32_t* __restrict__ MASK, int Ntok, int ld,
    float scale, float p_drop, int nheads) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * KP];
  __shared__ __attribute__((aligned(16))) float Gs[64 * KP];   // dO tile
  __shared__ float Ls[64], Ds[64];
  __shared__ uint32_t Mw[4][64];     // keep-bit word of (query, this wave's 32-key group)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const int key = qb_ * 128 + wave * 32 + li;

  float4 kf[4], vf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    kf[t] = *reinterpret_cast<const float4*>(K + base + (size_t)key * ld + 8 * t + 4 * lh);
    vf[t] = *reinterpret_cast<const float4*>(V + base + (size_t)key * ld + 8 * t + 4 * lh);
  }
  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)(uint32_t)(p_drop * 65536.0f + 0.5f) / 65536.f) : 1.f;

  float4 qreg0, qreg1, greg0, greg1;
  float lreg = 0.f, dreg = 0.f;
  uint32_t mreg = 0u;
  const int srow = tid >> 3, scol = (tid & 7) * 4;
#define LOAD_QTILE(qt)                                                                  \
  do {                                                                                  \
    size_t off0_ = base + (size_t)((qt) * 64 + srow) * ld + scol;                       \
    size_t off1_ = off0_ + (size_t)32 * ld;                                             \
    qreg0 = *reinterpret_cast<const float4*>(Q + off0_);                                \
    qreg1 = *reinterpret_cast<const float4*>(Q + off1_);                                \
    greg0 = *reinterpret_cast<const float4*>(dO + off0_);                               \
    greg1 = *reinterpret_cast<const float4*>(dO + off1_);                               \
    if (tid < 64) {                                                                     \
      lreg = LSE[sbase + (qt) * 64 + tid];                                              \
      dreg = Dv[sbase + (qt) * 64 + tid];                                               \
    }                                                                                   \
    if (DROPOUT)                                                                        \
      mreg = MASK[(sbase + (qt) * 64 + (tid & 63)) * (Ntok / 32) + qb_ * 4 + (tid >> 6)]; \
  } while (0)
  const int ntiles = Ntok / 64;
  LOAD_QTILE(0);
  for (int qt = 0; qt < ntiles; ++qt) {
    *reinterpret_cast<float4*>(&Qs[srow * KP + scol]) =
        make_float4(qreg0.x * scale, qreg0.y * scale, qreg0.z * scale, qreg0.w * scale);
    *reinterpret_cast<float4*>(&Qs[(srow + 32) * KP + scol]) =
        make_float4(qreg1.x * scale, qreg1.y * scale, qreg1.z * scale, qreg1.w * scale);
    *reinterpret_cast<float4*>(&Gs[srow * KP + scol]) = greg0;
    *reinterpret_cast<float4*>(&Gs[(srow + 32) * KP + scol]) = greg1;
    if (tid < 64) {
      Ls[tid] = lreg;
      Ds[tid] = dreg;
    }
    if (DROPOUT) Mw[tid >> 6][tid & 63] = mreg;
    __syncthreads();
    if (qt + 1 < ntiles) LOAD_QTILE(qt + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 a = *reinterpret_cast<const float4*>(&Qs[(sub * 32 + li) * KP + 8 * t + 4 * lh]);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, kf[t].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, kf[t].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, kf[t].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, kf[t].w, s, 0, 0, 0);
        float4 g = *reinterpret_cast<const float4*>(&Gs[(sub * 32 + li) * KP + 8 * t + 4 * lh]);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g.x, vf[t].x, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g.y, vf[t].y, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g.z, vf[t].z, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(g.w, vf[t].w, dp, 0, 0, 0);
      }
      // regs r <-> query sub*32 + key_of(r,lh); lane <-> key
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ql = sub * 32 + key_of(r, lh);
        float p = __expf(s[r] - Ls[ql]);
        float pd = p, dpe = dp[r];
        if (DROPOUT) {
          bool keep = (Mw[wave][ql] >> li) & 1u;
          pd = keep ? p * inv_keep : 0.f;
          dpe = keep ? dpe * inv_keep : 0.f;
        }
        s[r] = pd;                       // P (dropped) for dV
        dp[r] = p * (dpe - Ds[ql]);      // dS
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ql = sub * 32 + key_of(r, lh);
        float g = Gs[ql * KP + li];
        dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(g, s[r], dvacc, 0, 0, 0);
        float a = Qs[ql * KP + li];
        dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dp[r], dkacc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* dkrow = dK + base + (size_t)key * ld;
  float* dvrow = dV + base + (size_t)key * ld;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<float4*>(dkrow + 8 * g + 4 * lh) =
        make_float4(dkacc[4 * g], dkacc[4 * g + 1], dkacc[4 * g + 2], dkacc[4 * g + 3]);
    *reinterpret_cast<float4*>(dvrow + 8 * g + 4 * lh) =
        make_float4(dvacc[4 * g], dvacc[4 * g + 1], dvacc[4 * g + 2], dvacc[4 * g + 3]);
  }
}

// pass 2: dQ.  block = 128 queries (wave = 32), loop over 64-key tiles.
//   S^T [key][q] : A = K rows (LDS b128), B = Qs (regs)     lane = query, regs = keys
//   dP^T[key][q] : A = V rows (LDS b128), B = dO (regs)
//   dQ^T[d][q] = sum_key K[key][d] dS[q][key] : A = K[key(s)][d=li] (LDS b32), B = dS reg s
template <bool DROPOUT>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    float* __restrict__ dQ, const uint32_t* __restrict__ MASK, int Ntok, int ld, float scale, float p_drop,
    int nheads) {
  __shared__ __attribute__((aligned(16))) float Ks[64 * KP];
  __shared__ __attribute__((aligned(16))) float Vs[64 * KP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const int q = qb_ * 128 + wave * 32 + li;

  float4 qf[4], gf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float4 v = *reinterpret_cast<const float4*>(Q + base + (size_t)q * ld + 8 * t + 4 * lh);
    qf[t] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    gf[t] = *reinterpret_cast<const float4*>(dO + base + (size_t)q * ld + 8 * t + 4 * lh);
  }
  const float lse = LSE[sbase + q], dd = Dv[sbase + q];
  f32x16 dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)(uint32_t)(p_drop * 65536.0f + 0.5f) / 65536.f) : 1.f;
  const uint32_t* mrow = MASK + (sbase + q) * (size_t)(Ntok / 32);

  // staging registers (named scalars: arrays captured by a lambda end up in scratch)
  float4 kreg0, kreg1, vreg0, vreg1;
  const int srow = tid >> 3, scol = (tid & 7) * 4;      // rows srow and srow+32
#define LOAD_TILE(kt)                                                                   \
  do {                                                                                  \
    size_t off0_ = base + (size_t)((kt) * 64 + srow) * ld + scol;                       \
    size_t off1_ = off0_ + (size_t)32 * ld;                                             \
    kreg0 = *reinterpret_cast<const float4*>(K + off0_);                                \
    kreg1 = *reinterpret_cast<const float4*>(K + off1_);                                \
    vreg0 = *reinterpret_cast<const float4*>(V + off0_);                                \
    vreg1 = *reinterpret_cast<const float4*>(V + off1_);                                \
  } while (0)
  const int ntiles = Ntok / 64;
  LOAD_TILE(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    *reinterpret_cast<float4*>(&Ks[srow * KP + scol]) = kreg0;
    *reinterpret_cast<float4*>(&Ks[(srow + 32) * KP + scol]) = kreg1;
    *reinterpret_cast<float4*>(&Vs[srow * KP + scol]) = vreg0;
    *reinterpret_cast<float4*>(&Vs[(srow + 32) * KP + scol]) = vreg1;
    __syncthreads();
    if (kt + 1 < ntiles) LOAD_TILE(kt + 1);
    uint2 mw = make_uint2(0u, 0u);
    if (DROPOUT) mw = *reinterpret_cast<const uint2*>(mrow + kt * 2);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 a = *reinterpret_cast<const float4*>(&Ks[(sub * 32 + li) * KP + 8 * t + 4 * lh]);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[t].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[t].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[t].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[t].w, s, 0, 0, 0);
        float4 c = *reinterpret_cast<const float4*>(&Vs[(sub * 32 + li) * KP + 8 * t + 4 * lh]);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.x, gf[t].x, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.y, gf[t].y, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.z, gf[t].z, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c.w, gf[t].w, dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __expf(s[r] - lse);
        float dpe = dp[r];
        if (DROPOUT) {
          uint32_t w = sub ? mw.y : mw.x;
          dpe = ((w >> key_of(r, lh)) & 1u) ? dpe * inv_keep : 0.f;
        }
        s[r] = p * (dpe - dd);           // dS
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float a = Ks[(sub * 32 + key_of(r, lh)) * KP + li];
        dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], dqacc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* row = dQ + base + (size_t)q * ld;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(row + 8 * g + 4 * lh) =
        make_float4(dqacc[4 * g] * scale, dqacc[4 * g + 1] * scale, dqacc[4 * g + 2] * scale,
                    dqacc[4 * g + 3] * scale);
}

int focr_attn_fwd_bx3(const float* q, const float* k, const float* v, float* o, float* lse, uint32_t* mask,
                      int B, int H, int Ntok, int ld, float scale, float p_drop, uint64_t seed,
                      hipStream_t stream);
int focr_attn_bwd_bx3(const float* q, const float* k, const float* v, const float* d_o, const float* lse,
                      const float* dwork, const uint32_t* mask, float* dq, float* dk, float* dv, int B, int H,
                      int Ntok, int ld, float scale, float p_drop, hipStream_t stream);

extern "C" int focr_attention_fwd(const float* q, const float* k, const float* v, float* o,
                                  float* lse, uint32_t* mask, int B, int H, int Ntok, int ld, float scale,
                                  float p_drop, uint64_t seed, hipStream_t stream) {
  FOCR_CHECK_ARG(q && k && v && o && lse, "null pointer");
  FOCR_CHECK_ARG(p_drop <= 0.f || mask, "dropout needs the keep-bit buffer [B,H,Ntok,Ntok/32]");
  FOCR_CHECK_ARG(Ntok % 128 == 0 && ld >= H * 32 && ld % 4 == 0, "need Ntok%128==0, head dim 32");
  FOCR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "bad dropout probability");
  if (p_drop > 0.f) {
    long nwords = (long)B * H * Ntok * (Ntok / 32);
    long gsz = (nwords + 255) / 256;
    if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(attn_mask_kernel, dim3((int)gsz), 256, 0, stream, mask, nwords, Ntok / 32, p_drop, seed);
  }
  if (focr_get_precision() == 1) {
    focr_attn_fwd_bx3(q, k, v, o, lse, mask, B, H, Ntok, ld, scale, p_drop, seed, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  dim3 grid(B * H * (Ntok / 128));   // see attn_block_decode()
  if (p_drop > 0.f)
    hipLaunchKernelGGL((attn_fwd_kernel<true>), grid, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, scale, p_drop, seed, H);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<false>), grid, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, scale, p_drop, seed, H);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// dwork: B*H*Ntok floats of workspace (D = rowsum(dO*O))
extern "C" int focr_attention_bwd(const float* q, const float* k, const float* v, const float* o,
                                  const float* d_o, const float* lse, const uint32_t* mask, float* dq,
                                  float* dk, float* dv, float* dwork, int B, int H, int Ntok, int ld,
                                  float scale, float p_drop, hipStream_t stream) {
  FOCR_CHECK_ARG(q && k && v && o && d_o && lse && dq && dk && dv && dwork, "null pointer");
  FOCR_CHECK_ARG(p_drop <= 0.f || mask, "dropout needs the keep-bit buffer written by the forward");
  FOCR_CHECK_ARG(Ntok % 128 == 0 && ld >= H * 32 && ld % 4 == 0, "need Ntok%128==0, head dim 32");
  long total = (long)B * Ntok * H;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(cdiv(total, 256)), 256, 0, stream, o, d_o, dwork, Ntok, ld, H, total);
  FOCR_LAUNCH_CHECK();
  if (focr_get_precision() == 1) {
    focr_attn_bwd_bx3(q, k, v, d_o, lse, dwork, mask, dq, dk, dv, B, H, Ntok, ld, scale, p_drop, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  dim3 grid(B * H * (Ntok / 128));   // see attn_block_decode()
  if (p_drop > 0.f) {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<true>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dk, dv, mask, Ntok, ld, scale, p_drop, H);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<true>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dq, mask, Ntok, ld, scale, p_drop, H);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<false>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dk, dv, mask, Ntok, ld, scale, p_drop, H);
    hipLaunchKernelGGL((attn_bwd_dq_kernel<false>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dq, mask, Ntok, ld, scale, p_drop, H);
  }
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// Fused attention on the bf16 MFMA pipe with split operands ("bf16x3"), fp32 accumulate.
//
// Every fp32 operand x is split as x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) and each product
// a.b is evaluated as  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi  on v_mfma_f32_32x32x16_bf16 (the dropped
// a_lo.b_lo term is ~2^-16 relative).  tools/exp_split_precision.py shows the end-to-end SR error of
// this arithmetic equals plain fp32's (1.1e-4 vs 1.2e-4 of max, B=4 train mode) while plain bf16 is
// 3e-2 -- so it keeps the 1e-3 parity gate and runs at 16/3 = 5.3x the f32-MFMA rate.
//
// Same structure, operand orientation (swapped products, lane-local softmax state, P used directly
// as an MFMA operand) and dropout keep-bit format as attention.hip; what changes:
//   * K/Q/dO row tiles live in LDS as bf16 hi/lo (pitch 40 -> conflict-free ds_read_b128 fragments),
//   * operands consumed with the contraction index along the LANE's 8-element fragment but stored
//     row-major in HBM (V for PV, dO/Q for dV/dK, K for dQ) are written TRANSPOSED into LDS
//     (pitch 68 -> conflict-free ds_read_b64), two adjacent keys packed per 32-bit store,
//   * P / dS are split in registers: reg 8m+e of the S accumulator is k-slot e of MFMA step m.
#include "focr_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f
#define RP 40   // row-tile pitch (bf16): 80 B
#define TP 68   // transposed-tile pitch (bf16): 136 B

__device__ __forceinline__ int key_of_b(int s, int lh) { return (s & 3) + 8 * (s >> 2) + 4 * lh; }

__device__ __forceinline__ void split1(float x, __bf16& hi, __bf16& lo) {
  hi = (__bf16)x;
  lo = (__bf16)(x - (float)hi);
}
// split 8 consecutive accumulator registers into MFMA operand fragments (pairwise: focr_common.h focr_split2)
__device__ __forceinline__ void split_regs(const f32x16& s, int m, bf16x8& hi, bf16x8& lo) {
  const float v[8] = {s[8 * m], s[8 * m + 1], s[8 * m + 2], s[8 * m + 3], s[8 * m + 4], s[8 * m + 5], s[8 * m + 6],
                      s[8 * m + 7]};
  focr_split8(v, hi, lo);
}
__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) { focr_split4(v, hi, lo); }
__device__ __forceinline__ bf16x8 cat44(bf16x4 a, bf16x4 b) {
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
#define MFMA3(acc, ah, al, bh, bl)                                            \
  do {                                                                        \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);      \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);      \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      \
  } while (0)

// row-major staging of 2 rows x 4 columns held by one thread (rows 2*rp, 2*rp+1; cols c0..c0+3)
__device__ __forceinline__ void put_rows(__bf16* Th, __bf16* Tl, int rp, int c0, float4 r0, float4 r1) {
  bf16x4 h0, l0, h1, l1;
  split4(r0, h0, l0);
  split4(r1, h1, l1);
  *reinterpret_cast<bf16x4*>(&Th[(2 * rp) * RP + c0]) = h0;
  *reinterpret_cast<bf16x4*>(&Tl[(2 * rp) * RP + c0]) = l0;
  *reinterpret_cast<bf16x4*>(&Th[(2 * rp + 1) * RP + c0]) = h1;
  *reinterpret_cast<bf16x4*>(&Tl[(2 * rp + 1) * RP + c0]) = l1;
}
// transposed staging: T[c0+e][2*rp .. 2*rp+1] = (r0[e], r1[e])
__device__ __forceinline__ void put_cols(__bf16* Th, __bf16* Tl, int rp, int c0, float4 r0, float4 r1) {
  const float a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bf16x2 h, l;
    focr_split2(f32x2{a[e], b[e]}, h, l);
    *reinterpret_cast<bf16x2*>(&Th[(c0 + e) * TP + 2 * rp]) = h;
    *reinterpret_cast<bf16x2*>(&Tl[(c0 + e) * TP + 2 * rp]) = l;
  }
}
__device__ __forceinline__ float4 scale4(float4 v, float s) { return make_float4(v.x * s, v.y * s, v.z * s, v.w * s); }

// fragment of 8 consecutive columns of one global row, scaled, as hi/lo
__device__ __forceinline__ void row_frag(const float* p, float sc, bf16x8& hi, bf16x8& lo) {
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  bf16x4 h0, l0, h1, l1;
  split4(scale4(a, sc), h0, l0);
  split4(scale4(b, sc), h1, l1);
  hi = cat44(h0, h1);
  lo = cat44(l0, l1);
}

// ---- pre-split operand planes ("PL" kernel variants) -----------------------------------------------------------
// The producers of Q / K / V (the packed projection, csrc/fe_chain.hip fe_qkv_fwd_kernel) and of dO (fe_bwd_b_kernel)
// can write their results ALREADY split, as rows of 256 bf16 in which every group of four columns is stored as
// [hi x 4 | lo x 4] (x = hi + lo; 512 bytes per row, the bytes of the fp32 row: producers and consumers move 16 bytes per
// lane and instruction exactly as with fp32, no extra memory instructions), Q pre-multiplied by scale * log2(e), dO by
// 1 / P(keep).  The PL variants of the three big kernels then stage tiles by copying words (global -> LDS,
// transposition = one v_perm per word pair) and take their register fragments straight from memory: the per-block fp32 -> bf16 hi/lo split of every K / V / Q / dO tile (a quarter of the dK/dV
// pass's VALU work) is done once by the producer instead of by every consumer block.  Same values as the fp32 path
// (the split is the same arithmetic), except that the dropout scale rides on dO instead of V in the dK/dV pass.
// eight consecutive columns (from a multiple of 8) of one row: two 16-byte words [hi4 | lo4][hi4 | lo4] -> hi / lo fragments
__device__ __forceinline__ void pl_frag(const __bf16* p, bf16x8& hi, bf16x8& lo) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  const u4 a = *reinterpret_cast<const u4*>(p), b = *reinterpret_cast<const u4*>(p + 8);
  const u4 h = {a[0], a[1], b[0], b[1]}, l = {a[2], a[3], b[2], b[3]};
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// element offset of column c (multiple of 4) inside a 256-element row
#define PLC(c) (2 * (c))
// two rows (2 rp, 2 rp + 1) x 4 columns (c0 ..) of one plane, held as two 8-byte words
__device__ __forceinline__ void pl_put_rows(__bf16* T, int rp, int c0, uint2 r0, uint2 r1) {
  *reinterpret_cast<uint2*>(&T[(2 * rp) * RP + c0]) = r0;
  *reinterpret_cast<uint2*>(&T[(2 * rp + 1) * RP + c0]) = r1;
}
// transposed: T[c0 + e][2 rp .. 2 rp + 1] = (r0[e], r1[e])
__device__ __forceinline__ void pl_put_cols(__bf16* T, int rp, int c0, uint2 r0, uint2 r1) {
  uint32_t* t = reinterpret_cast<uint32_t*>(T);
  t[((c0 + 0) * TP + 2 * rp) >> 1] = __builtin_amdgcn_perm(r1.x, r0.x, 0x05040100u);
  t[((c0 + 1) * TP + 2 * rp) >> 1] = __builtin_amdgcn_perm(r1.x, r0.x, 0x07060302u);
  t[((c0 + 2) * TP + 2 * rp) >> 1] = __builtin_amdgcn_perm(r1.y, r0.y, 0x05040100u);
  t[((c0 + 3) * TP + 2 * rp) >> 1] = __builtin_amdgcn_perm(r1.y, r0.y, 0x07060302u);
}
__device__ __forceinline__ uint4 pl_ld(const __bf16* p) { return *reinterpret_cast<const uint4*>(p); }   // [hi4 | lo4]
__device__ __forceinline__ uint2 pl_hi(uint4 w) { return make_uint2(w.x, w.y); }
__device__ __forceinline__ uint2 pl_lo(uint4 w) { return make_uint2(w.z, w.w); }

// =======================================================================================
// forward
// =======================================================================================
template <bool DROPOUT>
__global__ __launch_bounds__(256, 4) void attn_fwd_bx3_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, float* __restrict__ O,
                                                           float* __restrict__ LSE, const uint32_t* __restrict__ MASK,
                                                           int Ntok, int ld, int ldo, float scale, float p_drop,
                                                           uint64_t seed, int nheads) {
  __shared__ __attribute__((aligned(16))) __bf16 Kh[64 * RP], Kl[64 * RP];
  __shared__ __attribute__((aligned(16))) __bf16 Vth[32 * TP], Vtl[32 * TP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const int q = qb_ * 128 + wave * 32 + li;

  bf16x8 qh[2], ql[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) row_frag(Q + base + (size_t)q * ld + 16 * m + 8 * lh, scale * LOG2E, qh[m], ql[m]);
  f32x16 oacc;   // scores are kept in log2 units (log2 e folded into the Q scale): p = exp2(s - m) is one v_exp_f32
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float mrun = -1e30f, l = 0.f;
  const uint32_t thr = DROPOUT ? attn_drop_thr16(p_drop) : 0u;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)thr / 65536.f) : 1.f;
  const int NG = Ntok / 32;
  const int qg = __builtin_amdgcn_readfirstlane(qb_ * 4 + wave);
  const uint64_t* mgrp = reinterpret_cast<const uint64_t*>(MASK) + ((size_t)bh_ * NG + qg) * NG * 16;

  const int rp = tid >> 3, c0 = (tid & 7) * 4;         // key pair, first of 4 d columns
  float4 k0, k1, v0, v1;
#define LOAD_KV(kt)                                                                   \
  do {                                                                                \
    size_t o0_ = base + (size_t)((kt) * 64 + 2 * rp) * ld + c0;                       \
    k0 = *reinterpret_cast<const float4*>(K + o0_);                                   \
    k1 = *reinterpret_cast<const float4*>(K + o0_ + ld);                              \
    v0 = *reinterpret_cast<const float4*>(V + o0_);                                   \
    v1 = *reinterpret_cast<const float4*>(V + o0_ + ld);                              \
  } while (0)
  const int ntiles = Ntok / 64;
  LOAD_KV(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    put_rows(Kh, Kl, rp, c0, k0, k1);
    put_cols(Vth, Vtl, rp, c0, v0, v1);
    __syncthreads();
    if (kt + 1 < ntiles) LOAD_KV(kt + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint64_t mk[16];
      if (DROPOUT) {
        const uint64_t* mp = mgrp + (size_t)(kt * 2 + sub) * 16;      // wave-uniform -> scalar loads
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = mp[r];
      }
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Kh[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
        bf16x8 al = *reinterpret_cast<const bf16x8*>(&Kl[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
        MFMA3(s, ah, al, qh[m], ql[m]);
      }
      float mx = s[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float mn = fmaxf(mrun, mx);
      float alpha = __builtin_amdgcn_exp2f(mrun - mn);
      mrun = mn;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = __builtin_amdgcn_exp2f(s[r] - mn);
        ls += p;
        s[r] = p;
      }
      if (DROPOUT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = keep_lanes(s[r], mk[r]);   // 1/(1-p) at the end
      }
      l = l * alpha + ls;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 ph, pl;
        split_regs(s, m, ph, pl);
        const int kc = sub * 32 + 16 * m + 4 * lh;
        bf16x8 vh = cat44(*reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc + 8]));
        bf16x8 vl = cat44(*reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc + 8]));
        MFMA3(oacc, vh, vl, ph, pl);
      }
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  float inv = inv_keep / l;
  float* orow = O + baseo + (size_t)q * ldo;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<float4*>(orow + 8 * g + 4 * lh) =
        make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
  if (lh == 0) LSE[(size_t)(b * H + h) * Ntok + q] = (mrun + __builtin_amdgcn_logf(l)) * LN2;   // natural log
}

#ifndef ATTN_RESCALE_THR
#define ATTN_RESCALE_THR 8.0f
#endif
// =======================================================================================
// forward, two query tiles per wave (block = 256 queries, wave = 2 x 32 queries)
//   Head dim 32 makes the softmax VALU work per score (max, exp2, sum, keep-select, hi/lo split: ~9 ops) as long as the
//   MFMA work per score (12 x 32 cycles per 32 x 32 tile); with one tile per wave the two run back to back inside
//   every wave (S MFMAs -> softmax -> PV MFMAs is one dependency chain).  With two independent tiles per wave the
//   softmax of tile A sits next to the S / PV MFMAs of tile B in the same instruction stream, every K / V fragment
//   read from LDS feeds two MFMA triples instead of one, and the per-block K/V staging is shared by 256 queries.
// =======================================================================================
__device__ __forceinline__ void hi_regs(const f32x16& s, int m, bf16x8& hi);
// PL: Q / K / V point to bf16 hi planes (row pitch ld = 128 elements, lo plane `pls` elements behind), Q pre-scaled
// MV (keep-word schedule, round 5): the 64-lane keep masks are wave-uniform scalar loads, and scalar loads share the
// lgkm counter with the LDS fragment reads -- out of order, so EVERY LDS wait behind them is a wait for them too.
//   0: masks requested at the top of a key group, in front of the K-fragment reads: the first score MFMA waits for the
//      HBM round trip of the keep words (the forward's 47 us dropout penalty);
//   1: K fragments are read and waited for FIRST, then the masks are requested and travel under the twelve score MFMAs,
//      the max tree and the exp2 stretch; the cross-half max exchange is a v_permlane32_swap (no LDS operation, no lgkm
//      wait) so that nothing between the request and the first v_cndmask waits on the counter.
template <bool DROPOUT, bool PL, int MV = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd2_bx3_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, float* __restrict__ O,
                                                            float* __restrict__ LSE, const uint32_t* __restrict__ MASK,
                                                            int Ntok, int ld, int ldo, float scale, float p_drop,
                                                            uint64_t seed, int nheads, long pls) {
  const __bf16* const Qp = reinterpret_cast<const __bf16*>(Q);
  const __bf16* const Kp = reinterpret_cast<const __bf16*>(K);
  const __bf16* const Vp = reinterpret_cast<const __bf16*>(V);
  __shared__ __attribute__((aligned(16))) __bf16 Kh[64 * RP], Kl[64 * RP];
  __shared__ __attribute__((aligned(16))) __bf16 Vth[32 * TP], Vtl[32 * TP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 256), Ntok / 256, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const size_t pbase = (size_t)b * Ntok * ld + h * 64;      // PL: ld = row pitch of the split rows in bf16 elements
  const int q0 = qb_ * 256 + wave * 64 + li;                 // tile t: query q0 + 32 t

  bf16x8 qh[2][2], ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (PL) {
        pl_frag(Qp + pbase + (size_t)(q0 + 32 * t) * ld + PLC(16 * m + 8 * lh), qh[t][m], ql[t][m]);
      } else {
        row_frag(Q + base + (size_t)(q0 + 32 * t) * ld + 16 * m + 8 * lh, scale * LOG2E, qh[t][m], ql[t][m]);
      }
    }
  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float mrun[2] = {MV == 2 ? 0.f : -1e30f, MV == 2 ? 0.f : -1e30f}, l[2] = {0.f, 0.f};
  const uint32_t thr = DROPOUT ? attn_drop_thr16(p_drop) : 0u;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)thr / 65536.f) : 1.f;
  const int NG = Ntok / 32;
  const int qg = __builtin_amdgcn_readfirstlane(qb_ * 8 + wave * 2);
  const uint64_t* mgrp = reinterpret_cast<const uint64_t*>(MASK) + ((size_t)bh_ * NG + qg) * NG * 16;

  const int rp = tid >> 3, c0 = (tid & 7) * 4;         // key pair, first of 4 d columns
  float4 k0, k1, v0, v1;
  uint4 pk[4];                                         // PL: K rows 2 rp, 2 rp + 1, V rows: [hi4 | lo4] each
#define PL_LOAD_KV(kt)                                                                \
  do {                                                                                \
    const size_t o0_ = pbase + (size_t)((kt) * 64 + 2 * rp) * ld + PLC(c0);          \
    pk[0] = pl_ld(Kp + o0_); pk[1] = pl_ld(Kp + o0_ + ld);                            \
    pk[2] = pl_ld(Vp + o0_); pk[3] = pl_ld(Vp + o0_ + ld);                            \
  } while (0)
  const int ntiles = Ntok / 64;
  if (PL) PL_LOAD_KV(0);
  else LOAD_KV(0);
  for (int kt = 0; kt < ntiles; ++kt) {
#ifdef ATTN_ABL_STAGE
    if (kt == 0)
#endif
    {
      if (PL) {
        pl_put_rows(Kh, rp, c0, pl_hi(pk[0]), pl_hi(pk[1]));
        pl_put_rows(Kl, rp, c0, pl_lo(pk[0]), pl_lo(pk[1]));
        pl_put_cols(Vth, rp, c0, pl_hi(pk[2]), pl_hi(pk[3]));
        pl_put_cols(Vtl, rp, c0, pl_lo(pk[2]), pl_lo(pk[3]));
      } else {
        put_rows(Kh, Kl, rp, c0, k0, k1);
        put_cols(Vth, Vtl, rp, c0, v0, v1);
      }
    }
    __syncthreads();
#ifndef ATTN_ABL_STAGE
    if (kt + 1 < ntiles) {
      if (PL) PL_LOAD_KV(kt + 1);
      else LOAD_KV(kt + 1);
    }
#endif
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // keep-bit lane masks of both tiles (wave-uniform -> scalar loads), requested before the score MFMAs so that
      // their latency is covered
      uint64_t mk[2][16];
      bf16x8 ah[2], al[2];
      if (MV >= 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          ah[m] = *reinterpret_cast<const bf16x8*>(&Kh[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
          al[m] = *reinterpret_cast<const bf16x8*>(&Kl[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
        }
        if (DROPOUT) {
          __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the fragments are here BEFORE the scalar requests go out
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (DROPOUT) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint64_t* mp = mgrp + ((size_t)t * NG + (kt * 2 + sub)) * 16;
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[t][r] = mp[r];
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the requests up here (the scheduler sinks them to their uses)
      }
      f32x16 s[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (MV < 1) {
          ah[m] = *reinterpret_cast<const bf16x8*>(&Kh[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
          al[m] = *reinterpret_cast<const bf16x8*>(&Kl[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
        }
#ifdef ATTN_ABL_MFMA_S
        s[0][0] += (float)ah[m][0] + (float)al[m][1]; s[1][0] += (float)ah[m][2];
#else
        MFMA3(s[0], ah[m], al[m], qh[0][m], ql[0][m]);
        MFMA3(s[1], ah[m], al[m], qh[1][m], ql[1][m]);
#endif
      }
#ifndef ATTN_ABL_SOFTMAX
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float ls = 0.f;
        if (MV == 2) {
          // Scores RELATIVE to the running reference first (the subtraction the exponent needs anyway): the max tree then
          // works on arithmetic results (no NaN-quieting v_max x, x in front of it) and every lane tests ITS OWN key half
          // against the threshold -- the ballot spans both halves of every query, so the wave takes the rescale branch
          // exactly when some query's full max exceeds it, and the cross-half exchange is only needed INSIDE that branch.
          // The first key group of a block always takes the branch (the reference starts at 0 and becomes the true max).
#pragma unroll
          for (int r = 0; r < 16; ++r) s[t][r] -= mrun[t];
          float mx = s[t][0];
#pragma unroll
          for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
          const bool first = kt == 0 && sub == 0;
          if (first || __builtin_amdgcn_ballot_w64(mx > ATTN_RESCALE_THR) != 0ull) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            const float delta = first ? mx : fmaxf(mx, 0.f);
            const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);       // (O and l are still zero on `first`)
            mrun[t] += delta;
            l[t] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] -= delta;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[t][r]);
            ls += p;
            s[t][r] = p;
          }
          l[t] += ls;
          continue;
        }
        float mx = s[t][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
        if (MV == 1) {
          // lanes l and l ^ 32 hold the two key halves of one query: swap the halves in the VALU (gfx950
          // v_permlane32_swap: upper row of the first operand <-> lower row of the second), max of the pair = max of
          // own and partner in every lane
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        } else {
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        }
        // Thresholded running max (wave-uniform branch): O / l are rescaled only when some query's max grew by more
        // than 2^ATTN_RESCALE_THR since the last rescale; otherwise P is merely bounded by 2^THR instead of 1 --
        // harmless for fp32 accumulation with split operands -- and the exp + 16 multiplies of the common case go away
        if (__builtin_amdgcn_ballot_w64(mx > mrun[t] + ATTN_RESCALE_THR) != 0ull) {
          const float mn = fmaxf(mrun[t], mx);
          const float alpha = __builtin_amdgcn_exp2f(mrun[t] - mn);
          mrun[t] = mn;
          l[t] *= alpha;
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
        }
        const float mn = mrun[t];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef ATTN_ABL_EXP
          const float p = s[t][r] - mn;
#else
          const float p = __builtin_amdgcn_exp2f(s[t][r] - mn);
#endif
          ls += p;
          s[t][r] = p;
        }
        if (DROPOUT && MV < 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[t][r] = keep_lanes(s[t][r], mk[t][r]);   // 1/(1-p) at the end
        }
        l[t] += ls;
      }
      if (DROPOUT && MV >= 1) {
        // both tiles' max / exp2 / row sums first (no mask needed), THEN the selects: the scalar requests have had the
        // score MFMAs and ~230 VALU instructions to arrive
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[t][r] = keep_lanes(s[t][r], mk[t][r]);   // 1/(1-p) at the end
      }
#endif
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int kc = sub * 32 + 16 * m + 4 * lh;
        bf16x8 vh = cat44(*reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc + 8]));
        bf16x8 vl = cat44(*reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc + 8]));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          bf16x8 ph, pl;
#ifdef ATTN_ABL_SPLIT
          hi_regs(s[t], m, ph);
          pl = ph;
#else
          split_regs(s[t], m, ph, pl);
#endif
#ifdef ATTN_ABL_MFMA_PV
          oacc[t][0] += (float)ph[0] + (float)pl[1] + (float)vh[0] + (float)vl[0];
#else
          MFMA3(oacc[t], vh, vl, ph, pl);
#endif
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float lt = l[t] + __shfl_xor(l[t], 32, 64);
    const float inv = inv_keep / lt;
    const int q = q0 + 32 * t;
    float* orow = O + baseo + (size_t)q * ldo;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(orow + 8 * g + 4 * lh) = make_float4(
          oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv, oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
    if (lh == 0) LSE[(size_t)(b * H + h) * Ntok + q] = (mrun[t] + __builtin_amdgcn_logf(lt)) * LN2;   // natural log
  }
}

// =======================================================================================
// forward, two query tiles per wave, score products ONE key group ahead (variant 2)
//   In attn_fwd2 the S MFMAs of a key group are followed immediately by the softmax that consumes them: the wave
//   waits out the matrix pipe, then the pipe idles through ~300 VALU instructions (ablation: removing the 12 S MFMAs
//   saves 121 us of 248, removing the 12 PV MFMAs only 64).  Here the scores of group g + 1 are issued BEFORE the
//   softmax of group g, so the softmax VALU work runs under independent MFMAs; K/V tiles are triple-buffered in LDS
//   so that the look-ahead may cross into the next 64-key tile with ONE barrier per tile.  The running-max rescale of
//   O / l is taken only when some query's max grew by more than 2^ATTN_RESCALE_THR (wave-uniform branch): P is then
//   bounded by 2^THR instead of 1, harmless in fp32 accumulation with split operands, and the 16 multiplies + exp of
//   the common no-change case disappear.
// =======================================================================================
#define F3_KB (64 * RP)
#define F3_VB (32 * TP)

__device__ __forceinline__ void f3_scores(f32x16 (&s)[2], const __bf16* Kh, const __bf16* Kl, int sub, int li, int lh,
                                          const bf16x8 (&qh)[2][2], const bf16x8 (&ql)[2][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Kh[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&Kl[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
    MFMA3(s[0], ah, al, qh[0][m], ql[0][m]);
    MFMA3(s[1], ah, al, qh[1][m], ql[1][m]);
  }
}
// running max of one tile + the (rare, wave-uniform) rescale of its O / l
__device__ __forceinline__ void f3_rowmax(const f32x16& s, f32x16& oacc, float& mrun, float& l) {
  float mx = s[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (__builtin_amdgcn_ballot_w64(mx > mrun + ATTN_RESCALE_THR) != 0ull) {     // wave-uniform
    const float mn = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f(mrun - mn);
    mrun = mn;
    l *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
  }
}
// ONE basic block: the 12 score MFMAs of the NEXT key group (sn) interleaved with the exponentials / sums / keep-bit
// selects of the CURRENT one (sc): 12 x 32 matrix-pipe cycles beside ~130 VALU + 32 transcendental issues
template <bool DROPOUT, bool NEXT>
__device__ __forceinline__ void f3_exp_scores(f32x16 (&sc)[2], f32x16 (&sn)[2], const float (&mrun)[2], float (&l)[2],
                                              const uint64_t* mp0, const uint64_t* mp1, const __bf16* Kh,
                                              const __bf16* Kl, int sub, int li, int lh, const bf16x8 (&qh)[2][2],
                                              const bf16x8 (&ql)[2][2]) {
  uint64_t mk[2][16];
  if (DROPOUT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { mk[0][r] = mp0[r]; mk[1][r] = mp1[r]; }
  }
  bf16x8 ah[2], al[2];
  if (NEXT) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      ah[m] = *reinterpret_cast<const bf16x8*>(&Kh[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
      al[m] = *reinterpret_cast<const bf16x8*>(&Kl[(sub * 32 + li) * RP + 16 * m + 8 * lh]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[t][r] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (NEXT) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      MFMA3(sn[0], ah[m], al[m], qh[0][m], ql[0][m]);
      MFMA3(sn[1], ah[m], al[m], qh[1][m], ql[1][m]);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __builtin_amdgcn_exp2f(sc[t][r] - mrun[t]);
      ls += p;
      if (DROPOUT) p = keep_lanes(p, mk[t][r]);      // 1/(1-p) at the end
      sc[t][r] = p;
    }
    l[t] += ls;
  }
  if (NEXT) {
    // MFMA = 0x8, VALU = 0x2, TRANS = 0x400
// same-accumulator MFMA triples stay back to back (an issue slot between two MFMAs on one accumulator costs ~43
    // cycles, MI355X_MICROARCH.md); the VALU / transcendental work goes between triples on different accumulators
#define F3_GROUP                                                   \
  __builtin_amdgcn_sched_group_barrier(0x8, 3, 0);                  \
  __builtin_amdgcn_sched_group_barrier(0x2, DROPOUT ? 24 : 18, 0);  \
  __builtin_amdgcn_sched_group_barrier(0x400, 8, 0);
    F3_GROUP F3_GROUP F3_GROUP F3_GROUP
#undef F3_GROUP
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void f3_pv(f32x16 (&oacc)[2], const f32x16 (&s)[2], const __bf16* Vth, const __bf16* Vtl,
                                      int sub, int li, int lh) {
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int kc = sub * 32 + 16 * m + 4 * lh;
    const bf16x8 vh = cat44(*reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc]),
                            *reinterpret_cast<const bf16x4*>(&Vth[li * TP + kc + 8]));
    const bf16x8 vl = cat44(*reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc]),
                            *reinterpret_cast<const bf16x4*>(&Vtl[li * TP + kc + 8]));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 ph, pl;
      split_regs(s[t], m, ph, pl);
      MFMA3(oacc[t], vh, vl, ph, pl);
    }
  }
}

template <bool DROPOUT>
__global__ __launch_bounds__(256, 2) void attn_fwd3_bx3_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, float* __restrict__ O,
                                                            float* __restrict__ LSE, const uint32_t* __restrict__ MASK,
                                                            int Ntok, int ld, int ldo, float scale, float p_drop,
                                                            uint64_t seed, int nheads) {
  __shared__ __attribute__((aligned(16))) __bf16 Kh[3 * F3_KB], Kl[3 * F3_KB];
  __shared__ __attribute__((aligned(16))) __bf16 Vth[3 * F3_VB], Vtl[3 * F3_VB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 256), Ntok / 256, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const int q0 = qb_ * 256 + wave * 64 + li;                 // tile t: query q0 + 32 t

  bf16x8 qh[2][2], ql[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m)
      row_frag(Q + base + (size_t)(q0 + 32 * t) * ld + 16 * m + 8 * lh, scale * LOG2E, qh[t][m], ql[t][m]);
  f32x16 oacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float mrun[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};
  const uint32_t thr = DROPOUT ? attn_drop_thr16(p_drop) : 0u;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)thr / 65536.f) : 1.f;
  const int NG = Ntok / 32;
  const int qg = __builtin_amdgcn_readfirstlane(qb_ * 8 + wave * 2);
  const uint64_t* mgrp = reinterpret_cast<const uint64_t*>(MASK) + ((size_t)bh_ * NG + qg) * NG * 16;

  const int rp = tid >> 3, c0 = (tid & 7) * 4;         // key pair, first of 4 d columns
  float4 k0, k1, v0, v1;
  const int ntiles = Ntok / 64;
  LOAD_KV(0);
  put_rows(Kh, Kl, rp, c0, k0, k1);
  put_cols(Vth, Vtl, rp, c0, v0, v1);
  __syncthreads();
  if (ntiles > 1) LOAD_KV(1);
  f32x16 sa[2], sb[2];
  f3_scores(sa, Kh, Kl, 0, li, lh, qh, ql);
  int buf = 0;
  for (int kt = 0; kt < ntiles; ++kt) {
    const __bf16 *kh = Kh + buf * F3_KB, *kl = Kl + buf * F3_KB, *vth = Vth + buf * F3_VB, *vtl = Vtl + buf * F3_VB;
    // ---- key group 0 of the tile: its scores (sa) were issued one group ago; group 1's go out under its softmax
    f3_rowmax(sa[0], oacc[0], mrun[0], l[0]);
    f3_rowmax(sa[1], oacc[1], mrun[1], l[1]);
    f3_exp_scores<DROPOUT, true>(sa, sb, mrun, l, mgrp + ((size_t)0 * NG + (kt * 2 + 0)) * 16,
                                 mgrp + ((size_t)1 * NG + (kt * 2 + 0)) * 16, kh, kl, 1, li, lh, qh, ql);
    f3_pv(oacc, sa, vth, vtl, 0, li, lh);
    // ---- key group 1: the next tile goes into the third buffer (the one nobody can still be reading), one barrier
    const int nbuf = buf == 2 ? 0 : buf + 1;
    f3_rowmax(sb[0], oacc[0], mrun[0], l[0]);
    f3_rowmax(sb[1], oacc[1], mrun[1], l[1]);
    if (kt + 1 < ntiles) {
      put_rows(Kh + nbuf * F3_KB, Kl + nbuf * F3_KB, rp, c0, k0, k1);
      put_cols(Vth + nbuf * F3_VB, Vtl + nbuf * F3_VB, rp, c0, v0, v1);
      __syncthreads();
      if (kt + 2 < ntiles) LOAD_KV(kt + 2);
      f3_exp_scores<DROPOUT, true>(sb, sa, mrun, l, mgrp + ((size_t)0 * NG + (kt * 2 + 1)) * 16,
                                   mgrp + ((size_t)1 * NG + (kt * 2 + 1)) * 16, Kh + nbuf * F3_KB, Kl + nbuf * F3_KB,
                                   0, li, lh, qh, ql);
    } else {
      f3_exp_scores<DROPOUT, false>(sb, sa, mrun, l, mgrp + ((size_t)0 * NG + (kt * 2 + 1)) * 16,
                                    mgrp + ((size_t)1 * NG + (kt * 2 + 1)) * 16, kh, kl, 0, li, lh, qh, ql);
    }
    f3_pv(oacc, sb, vth, vtl, 1, li, lh);
    buf = nbuf;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float lt = l[t] + __shfl_xor(l[t], 32, 64);
    const float inv = inv_keep / lt;
    const int q = q0 + 32 * t;
    float* orow = O + baseo + (size_t)q * ldo;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(orow + 8 * g + 4 * lh) = make_float4(
          oacc[t][4 * g] * inv, oacc[t][4 * g + 1] * inv, oacc[t][4 * g + 2] * inv, oacc[t][4 * g + 3] * inv);
    if (lh == 0) LSE[(size_t)(b * H + h) * Ntok + q] = (mrun[t] + __builtin_amdgcn_logf(lt)) * LN2;   // natural log
  }
}

// =======================================================================================
// backward pass 1: dK, dV   (block = 128 keys, wave = 32 keys, loop over 64-query tiles)
//   S[q][key]  : A = Qs rows (LDS), B = K (regs)        dP[q][key] : A = dO rows (LDS), B = V (regs)
//   dV^T[d][key] = sum_q dO[q][d] Pd[q][key] : A = dO^T (LDS transposed), B = split(Pd) regs
//   dK^T[d][key] = sum_q Qs[q][d] dS[q][key] : A = Qs^T (LDS transposed), B = split(dS) regs
// =======================================================================================
#define MFMA1(acc, ah, bh) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0)

// hi parts only (bf16 gradient accumulation, precision mode 2)
__device__ __forceinline__ void hi_regs(const f32x16& s, int m, bf16x8& hi) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const bf16x2 h = __builtin_convertvector(f32x2{s[8 * m + e], s[8 * m + e + 1]}, bf16x2);
    hi[e] = h[0];
    hi[e + 1] = h[1];
  }
}
// transposed staging of the hi plane only
__device__ __forceinline__ void put_cols_hi(__bf16* Th, int rp, int c0, float4 r0, float4 r1) {
  const float a[4] = {r0.x, r0.y, r0.z, r0.w}, b[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16x2 h = __builtin_convertvector(f32x2{a[e], b[e]}, bf16x2);
    *reinterpret_cast<bf16x2*>(&Th[(c0 + e) * TP + 2 * rp]) = h;
  }
}

// PL: Q (pre-scaled) / K / V / dO (pre-multiplied by 1 / P(keep)) are bf16 hi planes (pitch ld = ldo = 128, lo plane `pls`
// elements behind); dK / dV are written with row pitch ldg
template <bool DROPOUT, bool FAST, bool PL>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_bx3_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    float* __restrict__ dK, float* __restrict__ dV, const uint32_t* __restrict__ MASK, int Ntok, int ld, int ldo,
    float scale, float p_drop, int nheads, long pls, int ldg) {
  const __bf16* const Qp = reinterpret_cast<const __bf16*>(Q);
  const __bf16* const Kp = reinterpret_cast<const __bf16*>(K);
  const __bf16* const Vp = reinterpret_cast<const __bf16*>(V);
  const __bf16* const Gp = reinterpret_cast<const __bf16*>(dO);
  __shared__ __attribute__((aligned(16))) __bf16 Qh[64 * RP], Ql[64 * RP], Gh[64 * RP], Gl[64 * RP];
  __shared__ __attribute__((aligned(16))) __bf16 Qth[32 * TP], Qtl[32 * TP], Gth[32 * TP], Gtl[32 * TP];
  __shared__ float Ls[64], Ds[64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const size_t pbase = (size_t)b * Ntok * ld + h * 64;      // PL: ld / ldo = row pitches of the split Q|K|V / dO rows
  const size_t pgbase = (size_t)b * Ntok * ldo + h * 64;
  const int key = qb_ * 128 + wave * 32 + li;

  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)attn_drop_thr16(p_drop) / 65536.f) : 1.f;
  bf16x8 kh[2], kl[2], vh[2], vl[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (PL) {        // the dropout scale rides on the dO planes
      const size_t o_ = pbase + (size_t)key * ld + PLC(16 * m + 8 * lh);
      pl_frag(Kp + o_, kh[m], kl[m]);
      pl_frag(Vp + o_, vh[m], vl[m]);
    } else {
      row_frag(K + base + (size_t)key * ld + 16 * m + 8 * lh, 1.f, kh[m], kl[m]);
      // V carries the dropout scale 1/(1-p): dP' = dO (V/(1-p))^T is all the dS formula below needs of it
      row_frag(V + base + (size_t)key * ld + 16 * m + 8 * lh, inv_keep, vh[m], vl[m]);
    }
  }
  f32x16 dkacc, dvacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }

  const int rp = tid >> 3, c0 = (tid & 7) * 4;
  float4 q0, q1, g0, g1;
  float lreg = 0.f, dreg = 0.f;
  uint32_t mreg0 = 0u, mreg1 = 0u;     // this lane's key word for the two 32-query groups of the tile
  const int NG = Ntok / 32;
  const uint32_t* mkey = MASK + ((size_t)bh_ * NG * NG + (qb_ * 4 + wave)) * 32 + mask_slot(li);
#define LOAD_QG(qt)                                                                   \
  do {                                                                                \
    size_t o0_ = base + (size_t)((qt) * 64 + 2 * rp) * ld + c0;                       \
    q0 = *reinterpret_cast<const float4*>(Q + o0_);                                   \
    q1 = *reinterpret_cast<const float4*>(Q + o0_ + ld);                              \
    g0 = *reinterpret_cast<const float4*>(dO + baseo + (size_t)((qt) * 64 + 2 * rp) * ldo + c0);       \
    g1 = *reinterpret_cast<const float4*>(dO + baseo + (size_t)((qt) * 64 + 2 * rp + 1) * ldo + c0);   \
    if (tid < 64) {                                                                   \
      lreg = LSE[sbase + (qt) * 64 + tid];                                            \
      dreg = Dv[sbase + (qt) * 64 + tid];                                             \
    }                                                                                 \
    if (DROPOUT) {                                                                    \
      mreg0 = mkey[(size_t)((qt) * 2) * NG * 32];                                     \
      mreg1 = mkey[(size_t)((qt) * 2 + 1) * NG * 32];                                 \
    }                                                                                 \
  } while (0)
  const int ntiles = Ntok / 64;
  uint4 pq[4];                                          // PL: Q rows 2 rp, 2 rp + 1, dO rows: [hi4 | lo4] each
#define PL_LOAD_QG(qt)                                                                \
  do {                                                                                \
    const size_t o0_ = pbase + (size_t)((qt) * 64 + 2 * rp) * ld + PLC(c0);          \
    const size_t g0_ = pgbase + (size_t)((qt) * 64 + 2 * rp) * ldo + PLC(c0);        \
    pq[0] = pl_ld(Qp + o0_); pq[1] = pl_ld(Qp + o0_ + ld);                            \
    pq[2] = pl_ld(Gp + g0_); pq[3] = pl_ld(Gp + g0_ + ldo);                           \
    if (tid < 64) {                                                                   \
      lreg = LSE[sbase + (qt) * 64 + tid];                                            \
      dreg = Dv[sbase + (qt) * 64 + tid];                                             \
    }                                                                                 \
    if (DROPOUT) {                                                                    \
      mreg0 = mkey[(size_t)((qt) * 2) * NG * 32];                                     \
      mreg1 = mkey[(size_t)((qt) * 2 + 1) * NG * 32];                                 \
    }                                                                                 \
  } while (0)
  if (PL) PL_LOAD_QG(0);
  else LOAD_QG(0);
  int boff[16];                                    // keep-bit offsets of the 16 accumulator registers, opaque SGPRs (below)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    boff[r] = (r & 3) + 8 * (r >> 2);
    asm volatile("" : "+s"(boff[r]));
  }
  for (int qt = 0; qt < ntiles; ++qt) {
    const uint32_t mcur0 = mreg0 >> (4 * lh), mcur1 = mreg1 >> (4 * lh);
    if (PL) {
      pl_put_rows(Qh, rp, c0, pl_hi(pq[0]), pl_hi(pq[1]));
      pl_put_rows(Ql, rp, c0, pl_lo(pq[0]), pl_lo(pq[1]));
      pl_put_rows(Gh, rp, c0, pl_hi(pq[2]), pl_hi(pq[3]));
      pl_put_rows(Gl, rp, c0, pl_lo(pq[2]), pl_lo(pq[3]));
      pl_put_cols(Qth, rp, c0, pl_hi(pq[0]), pl_hi(pq[1]));
      pl_put_cols(Gth, rp, c0, pl_hi(pq[2]), pl_hi(pq[3]));
      if (!FAST) {
        pl_put_cols(Qtl, rp, c0, pl_lo(pq[0]), pl_lo(pq[1]));
        pl_put_cols(Gtl, rp, c0, pl_lo(pq[2]), pl_lo(pq[3]));
      }
    } else {
      q0 = scale4(q0, scale * LOG2E);     // log2 units: p = exp2(s - lse*log2e); dK is rescaled by ln2 at the end
      q1 = scale4(q1, scale * LOG2E);
      put_rows(Qh, Ql, rp, c0, q0, q1);
      put_rows(Gh, Gl, rp, c0, g0, g1);
      if (FAST) {
        put_cols_hi(Qth, rp, c0, q0, q1);
        put_cols_hi(Gth, rp, c0, g0, g1);
      } else {
        put_cols(Qth, Qtl, rp, c0, q0, q1);
        put_cols(Gth, Gtl, rp, c0, g0, g1);
      }
    }
    if (tid < 64) { Ls[tid] = -lreg * LOG2E; Ds[tid] = dreg; }     // -LSE: the score accumulators START there
    __syncthreads();
    if (qt + 1 < ntiles) {
      if (PL) PL_LOAD_QG(qt + 1);
      else LOAD_QG(qt + 1);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      // s starts at -LSE of its query row (register r <-> query key_of_b(r, lh)): the MFMAs deliver s - lse for free
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = Ls[sub * 32 + key_of_b(r, lh)]; dp[r] = 0.f; }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int off = (sub * 32 + li) * RP + 16 * m + 8 * lh;
        bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Qh[off]), al = *reinterpret_cast<const bf16x8*>(&Ql[off]);
        MFMA3(s, ah, al, kh[m], kl[m]);
        bf16x8 gh = *reinterpret_cast<const bf16x8*>(&Gh[off]), gl = *reinterpret_cast<const bf16x8*>(&Gl[off]);
        MFMA3(dp, gh, gl, vh[m], vl[m]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qlq = sub * 32 + key_of_b(r, lh);
        // dS = P (M dP' - D) = (M P) dP' - P D with M the keep mask: one AND instead of two, the rest an fma
        const float p = __builtin_amdgcn_exp2f(s[r]);
        float pd = p;
        if (DROPOUT) {
          // query bit of this lane's key word.  The bit offset comes from an SGPR the compiler cannot see through:
          // with a literal offset it rewrites sbfe + and as and + cmp + cndmask (3 VALU operations instead of 2)
          const int mk = bit_sext(sub ? mcur1 : mcur0, boff[r]);
          pd = __int_as_float(__float_as_int(p) & mk);              // 1/(1-p) folded into the dV store
        }
        s[r] = pd;
        dp[r] = fmaf(pd, dp[r], -p * Ds[qlq]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int qc = sub * 32 + 16 * m + 4 * lh;
        bf16x8 gh = cat44(*reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc]),
                          *reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc + 8]));
        bf16x8 ah = cat44(*reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc]),
                          *reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc + 8]));
        if (FAST) {
          bf16x8 ph, sh;
          hi_regs(s, m, ph);
          hi_regs(dp, m, sh);
          MFMA1(dvacc, gh, ph);
          MFMA1(dkacc, ah, sh);
        } else {
          bf16x8 ph, pl, sh, sl;
          split_regs(s, m, ph, pl);
          split_regs(dp, m, sh, sl);
          bf16x8 gl = cat44(*reinterpret_cast<const bf16x4*>(&Gtl[li * TP + qc]),
                            *reinterpret_cast<const bf16x4*>(&Gtl[li * TP + qc + 8]));
          MFMA3(dvacc, gh, gl, ph, pl);
          bf16x8 al = cat44(*reinterpret_cast<const bf16x4*>(&Qtl[li * TP + qc]),
                            *reinterpret_cast<const bf16x4*>(&Qtl[li * TP + qc + 8]));
          MFMA3(dkacc, ah, al, sh, sl);
        }
      }
    }
    __syncthreads();
  }
  const size_t gbase = (size_t)b * Ntok * ldg + h * 32;
  float* dkrow = dK + gbase + (size_t)key * ldg;
  float* dvrow = dV + gbase + (size_t)key * ldg;
  const float vsc = PL ? 1.f : inv_keep;               // PL: dO already carries 1 / P(keep)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *reinterpret_cast<float4*>(dkrow + 8 * g + 4 * lh) =
        make_float4(dkacc[4 * g] * LN2, dkacc[4 * g + 1] * LN2, dkacc[4 * g + 2] * LN2, dkacc[4 * g + 3] * LN2);
    *reinterpret_cast<float4*>(dvrow + 8 * g + 4 * lh) =
        make_float4(dvacc[4 * g] * vsc, dvacc[4 * g + 1] * vsc, dvacc[4 * g + 2] * vsc, dvacc[4 * g + 3] * vsc);
  }
}

// =======================================================================================
// backward pass 2: dQ   (block = 128 queries, wave = 32, loop over 64-key tiles)
//   S^T[key][q] : A = K rows (LDS), B = Qs (regs)     dP^T[key][q] : A = V rows (LDS), B = dO (regs)
//   dQ^T[d][q] = sum_key K[key][d] dS[q][key] : A = K^T (LDS transposed), B = split(dS) regs
// =======================================================================================
template <bool DROPOUT, bool FAST>
__global__ __launch_bounds__(256, 4) void attn_bwd_dq_bx3_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    float* __restrict__ dQ, const uint32_t* __restrict__ MASK, int Ntok, int ld, int ldo, float scale, float p_drop,
    int nheads) {
  __shared__ __attribute__((aligned(16))) __bf16 Kh[64 * RP], Kl[64 * RP], Vh[64 * RP], Vl[64 * RP];
  __shared__ __attribute__((aligned(16))) __bf16 Kth[32 * TP], Ktl[32 * TP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 128), Ntok / 128, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const int q = qb_ * 128 + wave * 32 + li;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)attn_drop_thr16(p_drop) / 65536.f) : 1.f;

  bf16x8 qh[2], ql[2], gh[2], gl[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    row_frag(Q + base + (size_t)q * ld + 16 * m + 8 * lh, scale * LOG2E, qh[m], ql[m]);
    // dO carries the dropout scale 1/(1-p) (dP' = V dO'^T); D was computed from the unscaled dO by the prep kernel
    row_frag(dO + baseo + (size_t)q * ldo + 16 * m + 8 * lh, inv_keep, gh[m], gl[m]);
  }
  const float lse = LSE[sbase + q] * LOG2E, dd = Dv[sbase + q];
  f32x16 dqacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;
  const int NG = Ntok / 32;
  const int qg = __builtin_amdgcn_readfirstlane(qb_ * 4 + wave);
  const uint64_t* mgrp = reinterpret_cast<const uint64_t*>(MASK) + ((size_t)bh_ * NG + qg) * NG * 16;

  const int rp = tid >> 3, c0 = (tid & 7) * 4;
  float4 k0, k1, v0, v1;
  const int ntiles = Ntok / 64;
  LOAD_KV(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    put_rows(Kh, Kl, rp, c0, k0, k1);
    if (FAST) put_cols_hi(Kth, rp, c0, k0, k1);
    else put_cols(Kth, Ktl, rp, c0, k0, k1);
    put_rows(Vh, Vl, rp, c0, v0, v1);
    __syncthreads();
    if (kt + 1 < ntiles) LOAD_KV(kt + 1);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint64_t mk[16];
      if (DROPOUT) {
        const uint64_t* mp = mgrp + (size_t)(kt * 2 + sub) * 16;      // wave-uniform -> scalar loads
#pragma unroll
        for (int r = 0; r < 16; ++r) mk[r] = mp[r];
      }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int off = (sub * 32 + li) * RP + 16 * m + 8 * lh;
        bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Kh[off]), al = *reinterpret_cast<const bf16x8*>(&Kl[off]);
        MFMA3(s, ah, al, qh[m], ql[m]);
        bf16x8 ch = *reinterpret_cast<const bf16x8*>(&Vh[off]), cl = *reinterpret_cast<const bf16x8*>(&Vl[off]);
        MFMA3(dp, ch, cl, gh[m], gl[m]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] - lse);
        float dpe = dp[r];
        if (DROPOUT) dpe = keep_lanes(dpe, mk[r]);
        s[r] = p * (dpe - dd);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int kc = sub * 32 + 16 * m + 4 * lh;
        bf16x8 ah = cat44(*reinterpret_cast<const bf16x4*>(&Kth[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Kth[li * TP + kc + 8]));
        if (FAST) {
          bf16x8 sh;
          hi_regs(s, m, sh);
          MFMA1(dqacc, ah, sh);
        } else {
          bf16x8 sh, sl;
          split_regs(s, m, sh, sl);
          bf16x8 al = cat44(*reinterpret_cast<const bf16x4*>(&Ktl[li * TP + kc]),
                            *reinterpret_cast<const bf16x4*>(&Ktl[li * TP + kc + 8]));
          MFMA3(dqacc, ah, al, sh, sl);
        }
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

// dQ with two query tiles per wave (block = 256 queries): the K / V staging of a 64-key tile (global loads, hi/lo split,
// transposed copy: ~75 VALU instructions per thread) and every K / V fragment read serve 256 queries instead of 128.
template <bool DROPOUT, bool FAST, bool PL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq2_bx3_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv,
    float* __restrict__ dQ, const uint32_t* __restrict__ MASK, int Ntok, int ld, int ldo, float scale, float p_drop,
    int nheads, long pls, int ldg) {
  const __bf16* const Qp = reinterpret_cast<const __bf16*>(Q);
  const __bf16* const Kp = reinterpret_cast<const __bf16*>(K);
  const __bf16* const Vp = reinterpret_cast<const __bf16*>(V);
  const __bf16* const Gp = reinterpret_cast<const __bf16*>(dO);
  __shared__ __attribute__((aligned(16))) __bf16 Kh[64 * RP], Kl[64 * RP], Vh[64 * RP], Vl[64 * RP];
  __shared__ __attribute__((aligned(16))) __bf16 Kth[32 * TP], Ktl[32 * TP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int bh_, qb_;
  attn_block_decode(blockIdx.x, gridDim.x / (Ntok / 256), Ntok / 256, bh_, qb_);
  const int H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const size_t pbase = (size_t)b * Ntok * ld + h * 64;      // PL: ld / ldo = row pitches of the split Q|K|V / dO rows
  const size_t pgbase = (size_t)b * Ntok * ldo + h * 64;
  const int q0 = qb_ * 256 + wave * 64 + li;                 // tile t: query q0 + 32 t
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)attn_drop_thr16(p_drop) / 65536.f) : 1.f;

  bf16x8 qh[2][2], ql[2][2], gh[2][2], gl[2][2];
  float lse[2], dd[2];
  f32x16 dqacc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int q = q0 + 32 * t;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      if (PL) {
        pl_frag(Qp + pbase + (size_t)q * ld + PLC(16 * m + 8 * lh), qh[t][m], ql[t][m]);
        pl_frag(Gp + pgbase + (size_t)q * ldo + PLC(16 * m + 8 * lh), gh[t][m], gl[t][m]);
      } else {
        row_frag(Q + base + (size_t)q * ld + 16 * m + 8 * lh, scale * LOG2E, qh[t][m], ql[t][m]);
        // dO carries the dropout scale 1/(1-p) (dP' = V dO'^T); D was computed from the unscaled dO by the prep kernel
        row_frag(dO + baseo + (size_t)q * ldo + 16 * m + 8 * lh, inv_keep, gh[t][m], gl[t][m]);
      }
    }
    lse[t] = LSE[sbase + q] * LOG2E;
    dd[t] = Dv[sbase + q];
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[t][r] = 0.f;
  }
  const int NG = Ntok / 32;
  const int qg = __builtin_amdgcn_readfirstlane(qb_ * 8 + wave * 2);
  const uint64_t* mgrp = reinterpret_cast<const uint64_t*>(MASK) + ((size_t)bh_ * NG + qg) * NG * 16;

  const int rp = tid >> 3, c0 = (tid & 7) * 4;
  float4 k0, k1, v0, v1;
  uint4 pk[4];                                         // PL: K rows 2 rp, 2 rp + 1, V rows: [hi4 | lo4] each
  const int ntiles = Ntok / 64;
  if (PL) PL_LOAD_KV(0);
  else LOAD_KV(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    if (PL) {
      pl_put_rows(Kh, rp, c0, pl_hi(pk[0]), pl_hi(pk[1]));
      pl_put_rows(Kl, rp, c0, pl_lo(pk[0]), pl_lo(pk[1]));
      pl_put_cols(Kth, rp, c0, pl_hi(pk[0]), pl_hi(pk[1]));
      if (!FAST) pl_put_cols(Ktl, rp, c0, pl_lo(pk[0]), pl_lo(pk[1]));
      pl_put_rows(Vh, rp, c0, pl_hi(pk[2]), pl_hi(pk[3]));
      pl_put_rows(Vl, rp, c0, pl_lo(pk[2]), pl_lo(pk[3]));
    } else {
      put_rows(Kh, Kl, rp, c0, k0, k1);
      if (FAST) put_cols_hi(Kth, rp, c0, k0, k1);
      else put_cols(Kth, Ktl, rp, c0, k0, k1);
      put_rows(Vh, Vl, rp, c0, v0, v1);
    }
    __syncthreads();
    if (kt + 1 < ntiles) {
      if (PL) PL_LOAD_KV(kt + 1);
      else LOAD_KV(kt + 1);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      uint64_t mk[2][16];
      if (DROPOUT) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const uint64_t* mp = mgrp + ((size_t)t * NG + (kt * 2 + sub)) * 16;      // wave-uniform -> scalar loads
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[t][r] = mp[r];
        }
        __builtin_amdgcn_sched_barrier(0);       // keep the requests up here (the scheduler sinks them to their uses)
      }
      // the score accumulators START at -LSE of the lane's query: the MFMAs deliver s - lse for free.  (Starting dP at -D
      // as well and selecting -D for dropped scores saves the subtraction below but costs 80 spilled VGPRs: 615 -> 667 us
      // for the backward pair, tools/gpu/r03_call27.sh.)
      f32x16 s[2], dp[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[t][r] = FAST ? -lse[t] : 0.f; dp[t][r] = 0.f; }   // (FAST only: registers)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int off = (sub * 32 + li) * RP + 16 * m + 8 * lh;
        bf16x8 ah = *reinterpret_cast<const bf16x8*>(&Kh[off]), al = *reinterpret_cast<const bf16x8*>(&Kl[off]);
        MFMA3(s[0], ah, al, qh[0][m], ql[0][m]);
        MFMA3(s[1], ah, al, qh[1][m], ql[1][m]);
        bf16x8 ch = *reinterpret_cast<const bf16x8*>(&Vh[off]), cl = *reinterpret_cast<const bf16x8*>(&Vl[off]);
        MFMA3(dp[0], ch, cl, gh[0][m], gl[0][m]);
        MFMA3(dp[1], ch, cl, gh[1][m], gl[1][m]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(FAST ? s[t][r] : s[t][r] - lse[t]);
          float dpe = dp[t][r];
          if (DROPOUT) dpe = keep_lanes(dpe, mk[t][r]);
          s[t][r] = p * (dpe - dd[t]);
        }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int kc = sub * 32 + 16 * m + 4 * lh;
        bf16x8 ah = cat44(*reinterpret_cast<const bf16x4*>(&Kth[li * TP + kc]),
                          *reinterpret_cast<const bf16x4*>(&Kth[li * TP + kc + 8]));
        if (FAST) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            bf16x8 sh;
            hi_regs(s[t], m, sh);
            MFMA1(dqacc[t], ah, sh);
          }
        } else {
          bf16x8 al = cat44(*reinterpret_cast<const bf16x4*>(&Ktl[li * TP + kc]),
                            *reinterpret_cast<const bf16x4*>(&Ktl[li * TP + kc + 8]));
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            bf16x8 sh, sl;
            split_regs(s[t], m, sh, sl);
            MFMA3(dqacc[t], ah, al, sh, sl);
          }
        }
      }
    }
    __syncthreads();
  }
  const size_t gbase = (size_t)b * Ntok * ldg + h * 32;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float* row = dQ + gbase + (size_t)(q0 + 32 * t) * ldg;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(row + 8 * g + 4 * lh) =
          make_float4(dqacc[t][4 * g] * scale, dqacc[t][4 * g + 1] * scale, dqacc[t][4 * g + 2] * scale,
                      dqacc[t][4 * g + 3] * scale);
  }
}

#include "attention_bwd1_bx3.h"
#ifdef FOCR_UBENCH_BWD1W          // one-wave-per-SIMD experiment (measured slower): tools/ubench only
#include "../../tools/ubench/attention_bwd1w_bx3.h"
#endif

// launchers used by the dispatching C ABI entry points in attention.hip
// 0: one query tile per wave (128-query blocks); 1: two tiles per wave (256-query blocks, needs Ntok % 256 == 0)
int focr_attn_fwd_bx3(const float* q, const float* k, const float* v, float* o, float* lse, uint32_t* mask,
                      int B, int H, int Ntok, int ld, int ldo, float scale, float p_drop, uint64_t seed,
                      hipStream_t stream) {
  if (focr_get_tuning(FOCR_TUNE_ATTN_FWD_VARIANT) == 2 && Ntok % 256 == 0) {
    dim3 grid2(B * H * (Ntok / 256));
    if (p_drop > 0.f)
      hipLaunchKernelGGL((attn_fwd3_bx3_kernel<true>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H);
    else
      hipLaunchKernelGGL((attn_fwd3_bx3_kernel<false>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H);
    return 0;
  }
  if (focr_get_tuning(FOCR_TUNE_ATTN_FWD_VARIANT) == 1 && Ntok % 256 == 0) {
    dim3 grid2(B * H * (Ntok / 256));
    const int mv = focr_get_tuning(FOCR_TUNE_ATTN_FWD_MASK);
    if (p_drop > 0.f && mv == 2)
      hipLaunchKernelGGL((attn_fwd2_bx3_kernel<true, false, 2>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H, 0L);
    else if (p_drop > 0.f && mv == 1)
      hipLaunchKernelGGL((attn_fwd2_bx3_kernel<true, false, 1>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H, 0L);
    else if (p_drop > 0.f)
      hipLaunchKernelGGL((attn_fwd2_bx3_kernel<true, false>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H, 0L);
    else if (mv == 2)
      hipLaunchKernelGGL((attn_fwd2_bx3_kernel<false, false, 2>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H, 0L);
    else
      hipLaunchKernelGGL((attn_fwd2_bx3_kernel<false, false>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo,
                         scale, p_drop, seed, H, 0L);
    return 0;
  }
  dim3 grid(B * H * (Ntok / 128));
  if (p_drop > 0.f)
    hipLaunchKernelGGL((attn_fwd_bx3_kernel<true>), grid, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo, scale,
                       p_drop, seed, H);
  else
    hipLaunchKernelGGL((attn_fwd_bx3_kernel<false>), grid, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ld, ldo, scale,
                       p_drop, seed, H);
  return 0;
}
int focr_attn_bwd_bx3(const float* q, const float* k, const float* v, const float* d_o, const float* lse,
                      const float* dwork, const uint32_t* mask, float* dq, float* dk, float* dv, int B, int H,
                      int Ntok, int ld, int ldo, float scale, float p_drop, hipStream_t stream) {
  const bool fast = focr_get_precision() >= 2;
#ifdef FOCR_UBENCH_BWD1W
  if (fast && focr_get_tuning(FOCR_TUNE_ATTN_BWD_DQ_VARIANT) == 3 && Ntok % 256 == 0) {
    // experiment: the single pass with one wave per SIMD (tools/ubench/attention_bwd1w_bx3.h)
    if (p_drop > 0.f) {
      (void)hipFuncSetAttribute((const void*)attn_bwd1w_bx3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS_BYTES);
      hipLaunchKernelGGL((attn_bwd1w_bx3_kernel<true>), dim3(B * H), 256, B1_LDS_BYTES, stream, q, k, v, d_o, lse, dwork, dq,
                         dk, dv, mask, Ntok, ld, ldo, ld, scale, p_drop, H);
    } else {
      (void)hipFuncSetAttribute((const void*)attn_bwd1w_bx3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS_BYTES);
      hipLaunchKernelGGL((attn_bwd1w_bx3_kernel<false>), dim3(B * H), 256, B1_LDS_BYTES, stream, q, k, v, d_o, lse, dwork, dq,
                         dk, dv, mask, Ntok, ld, ldo, ld, scale, p_drop, H);
    }
    return 0;
  }
#endif
  // Variant (tuning key 3): 4 (default) = by grid size.  The single pass launches ONE block per (batch, head) that walks all
  // key chunks and query tiles: 157 us whatever the batch up to B * H = 128 blocks (a quarter of the CUs at the reference
  // README's batch 16), where the two-pass kernels (one block per 128 / 256 queries) take 78 / 101 / 137 / 163 us at
  // B = 8 / 16 / 24 / 32 -- and 611 against 412 at B = 128 (profiles/r06_attn_bwd_batch.txt).  2 = single pass always.
  const int variant = focr_get_tuning(FOCR_TUNE_ATTN_BWD_DQ_VARIANT);
  if (fast && (variant == 2 || (variant == 4 && B * H >= 128)) && Ntok % 256 == 0) {
    // single pass: dQ, dK, dV from one S / dP evaluation (attention_bwd1_bx3.h); 138.5 KB of LDS per block.  The
    // attribute is per device: set on every call (a host-side table lookup) rather than cached in a process-wide flag.
    // Mode 3 (bf16 data gradients): dP = dO V^T as a single bf16 product (template flag DP1).
#ifdef B1_NO_DP1                     // A/B builds (tools/gpu): the split dP product in every mode
    const bool dp1 = false;
#else
    const bool dp1 = focr_get_precision() >= 3;
#endif
    // 138.5 KB of dynamic LDS: the attribute is set once per device and instantiation (focr_dev_flags); a device that
    // refuses it falls through to the two-pass kernels below (static LDS only) instead of failing the launch
    bool launched = false;
#define LAUNCH_BWD1(DR, D1)                                                                                       \
  do {                                                                                                            \
    static focr_dev_flags attr_set_;                                                                              \
    bool ok_ = true;                                                                                              \
    if (focr_dev_first(attr_set_)) {                                                                              \
      ok_ = hipFuncSetAttribute((const void*)attn_bwd1_bx3_kernel<DR, D1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                B1_LDS_BYTES) == hipSuccess;                                                      \
      if (ok_) focr_dev_mark(attr_set_);                                                                          \
      else (void)hipGetLastError();                                                                               \
    }                                                                                                             \
    if (ok_) {                                                                                                    \
      hipLaunchKernelGGL((attn_bwd1_bx3_kernel<DR, D1>), dim3(B * H), 512, B1_LDS_BYTES, stream, q, k, v, d_o, lse, dwork, \
                         dq, dk, dv, mask, Ntok, ld, ldo, ld, scale, p_drop, H);                                  \
      launched = true;                                                                                            \
    }                                                                                                             \
  } while (0)
    if (p_drop > 0.f) {
      if (dp1) LAUNCH_BWD1(true, true);
      else LAUNCH_BWD1(true, false);
    } else {
      if (dp1) LAUNCH_BWD1(false, true);
      else LAUNCH_BWD1(false, false);
    }
#undef LAUNCH_BWD1
    if (launched) return 0;
  }
  dim3 grid(B * H * (Ntok / 128));
  const bool dq2 = variant != 0 && Ntok % 256 == 0;      // two query tiles per wave in the dQ pass
#define LAUNCH_BWD(DR, FA)                                                                                        \
  do {                                                                                                            \
    hipLaunchKernelGGL((attn_bwd_dkv_bx3_kernel<DR, FA, false>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dk, dv, \
                       mask, Ntok, ld, ldo, scale, p_drop, H, 0L, ld);                                            \
    if (dq2)                                                                                                      \
      hipLaunchKernelGGL((attn_bwd_dq2_bx3_kernel<DR, FA, false>), dim3(B * H * (Ntok / 256)), 256, 0, stream, q, k, v, d_o, \
                         lse, dwork, dq, mask, Ntok, ld, ldo, scale, p_drop, H, 0L, ld);                          \
    else                                                                                                          \
      hipLaunchKernelGGL((attn_bwd_dq_bx3_kernel<DR, FA>), grid, 256, 0, stream, q, k, v, d_o, lse, dwork, dq, mask, \
                         Ntok, ld, ldo, scale, p_drop, H);                                                        \
  } while (0)
  if (p_drop > 0.f) {
    if (fast) LAUNCH_BWD(true, true);
    else LAUNCH_BWD(true, false);
  } else {
    if (fast) LAUNCH_BWD(false, true);
    else LAUNCH_BWD(false, false);
  }
  return 0;
}

#ifdef FOCR_UBENCH_PLANES          // tools/ubench only (see attention.hip)
// ---------------------------------------------------------------------------------------------------------------
// pre-split operand planes: producer for arbitrary fp32 inputs + the launchers of the PL kernel variants
// ---------------------------------------------------------------------------------------------------------------
// x [rows][ld] (128 columns used) -> rows of 256 bf16, every 4 columns as [hi4 | lo4]; values multiplied by `mul`
__global__ __launch_bounds__(256) void attn_make_planes_kernel(const float* __restrict__ x, __bf16* __restrict__ out,
                                                               long rows, int ld, float mul) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // one float4 each
  if (i >= rows * 32) return;
  const long r = i >> 5;
  const int c = (int)(i & 31) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
  const float a[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
  bf16x4 h, l;
  split4(make_float4(a[0], a[1], a[2], a[3]), h, l);
  *reinterpret_cast<bf16x8*>(out + r * 256 + 2 * c) = cat44(h, l);
}
int focr_attn_make_planes(const float* x, void* planes, long rows, int ld, float mul, hipStream_t stream) {
  hipLaunchKernelGGL(attn_make_planes_kernel, dim3((unsigned)((rows * 32 + 255) / 256)), 256, 0, stream, x,
                     reinterpret_cast<__bf16*>(planes), rows, ld, mul);
  return 0;
}
// qp / kp / vp (/ gp): [B * Ntok][256] bf16 split rows ([hi4 | lo4] per 4 columns).  Ntok % 256 == 0, H = 4.
int focr_attn_fwd_bx3_planes(const void* qp, const void* kp, const void* vp, float* o, float* lse, const uint32_t* mask,
                             int B, int H, int Ntok, int ldp, int ldo, float p_drop, hipStream_t stream) {
  const long pls = 0;
  const float* q = reinterpret_cast<const float*>(qp);
  const float* k = reinterpret_cast<const float*>(kp);
  const float* v = reinterpret_cast<const float*>(vp);
  dim3 grid2(B * H * (Ntok / 256));
  if (p_drop > 0.f && focr_get_tuning(FOCR_TUNE_ATTN_FWD_MASK) == 1)
    hipLaunchKernelGGL((attn_fwd2_bx3_kernel<true, true, 1>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ldp, ldo,
                       1.f, p_drop, (uint64_t)0, H, pls);
  else if (p_drop > 0.f)
    hipLaunchKernelGGL((attn_fwd2_bx3_kernel<true, true>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ldp, ldo,
                       1.f, p_drop, (uint64_t)0, H, pls);
  else
    hipLaunchKernelGGL((attn_fwd2_bx3_kernel<false, true>), grid2, 256, 0, stream, q, k, v, o, lse, mask, Ntok, ldp, ldo,
                       1.f, p_drop, (uint64_t)0, H, pls);
  return 0;
}
int focr_attn_bwd_bx3_planes(const void* qp, const void* kp, const void* vp, const void* gp, const float* lse,
                             const float* dwork, const uint32_t* mask, float* dq, float* dk, float* dv, int B, int H,
                             int Ntok, int ldp, int ldgp, int ldg, float scale, float p_drop, hipStream_t stream) {
  const long pls = 0;
  const float* q = reinterpret_cast<const float*>(qp);
  const float* k = reinterpret_cast<const float*>(kp);
  const float* v = reinterpret_cast<const float*>(vp);
  const float* g = reinterpret_cast<const float*>(gp);
  const bool fast = focr_get_precision() >= 2;
#define LAUNCH_BWD_PL(DR, FA)                                                                                      \
  do {                                                                                                             \
    hipLaunchKernelGGL((attn_bwd_dkv_bx3_kernel<DR, FA, true>), dim3(B * H * (Ntok / 128)), 256, 0, stream, q, k, v, g, \
                       lse, dwork, dk, dv, mask, Ntok, ldp, ldgp, scale, p_drop, H, pls, ldg);                     \
    hipLaunchKernelGGL((attn_bwd_dq2_bx3_kernel<DR, FA, true>), dim3(B * H * (Ntok / 256)), 256, 0, stream, q, k, v, g, \
                       lse, dwork, dq, mask, Ntok, ldp, ldgp, scale, p_drop, H, pls, ldg);                         \
  } while (0)
  if (p_drop > 0.f) {
    if (fast) LAUNCH_BWD_PL(true, true);
    else LAUNCH_BWD_PL(true, false);
  } else {
    if (fast) LAUNCH_BWD_PL(false, true);
    else LAUNCH_BWD_PL(false, false);
  }
  return 0;
}
#endif

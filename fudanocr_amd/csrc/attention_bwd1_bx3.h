// Single-pass attention backward (round 4): dQ, dK and dV from ONE evaluation of S, P, dP and dS per (query tile, key tile).
// Included by attention_bx3.hip (uses its staging / split helpers); precision modes 2 / 3 (bf16 gradient accumulation).
//
// The two-pass backward (attn_bwd_dkv + attn_bwd_dq2 above) evaluates the score tile, exp2, the keep-bit select, the
// hi/lo split and dP = dO V^T twice, because an MFMA result C[M][N] (lanes = N, registers = M) can only be LEFT-multiplied
// by the next MFMA (it is a B operand with k = M): S[q][key] feeds the sums over queries (dK^T, dV^T), S^T[key][q] the
// sums over keys (dQ^T), and the passes compute one orientation each: 30 MFMAs per 32 x 32 tile.  Here:
//   * one block owns a (batch, head); a wave owns 32 keys of the current 256-key chunk (K / V fragments and the dK / dV
//     accumulators in registers, exactly as in the dK/dV pass) and walks over the 64-query tiles staged in LDS;
//   * dS (bf16, the precision the accumulation uses anyway) is written to LDS as T[key][q] -- 4 adjacent queries per
//     lane and store, XOR-swizzled 8-byte chunks -- and read back TRANSPOSED by ds_read_b64_tr_b16 (gfx950's transpose
//     read: within 16 lanes, lane l receives element l & 3 of the 8-byte pieces addressed by lanes 4 j + (l >> 2)) as the
//     B operand dS^T[key][q] of  dQ^T[d][q] += K^T[d][key] dS^T[key][q]  on v_mfma_f32_16x16x32_bf16: every wave reduces
//     one 16-query x 16-column output tile over ALL 256 keys of the chunk -- the cross-wave sum over keys happens in the
//     MFMA's k dimension, on chip, no atomics;
//   * dQ accumulates across the Ntok / 256 chunks in its own output rows (each lane re-reads the float4 it wrote one
//     chunk earlier: no cross-thread communication through global memory);
//   * 18 MFMA-equivalents per tile (S 6, dP 6, dV 2, dK 2, dQ 2) and ONE exp / keep-bit / dS evaluation.
// One barrier per query tile: the staged Q / dO tiles and T are double-buffered.  LDS 141 824 B: one 8-wave block per CU.

#include <type_traits>
#define B1_STAGE 29184                       // Qh Ql Gh Gl [64][RP] + Qth Gth [32][TP]
#define B1_OFF_LS (2 * B1_STAGE)             // float Ls[2][64], Ds[2][64]
#define B1_OFF_T (B1_OFF_LS + 1024)          // 2 x [256 keys][64 q] bf16, 128-byte rows
#define B1_T_BYTES 32768
#define B1_OFF_KT (B1_OFF_T + 2 * B1_T_BYTES)
#define B1_KT_PITCH 528                      // 256 keys x 2 B + 16: conflict-free ds_read_b128 of the A fragments
#define B1_LDS_BYTES (B1_OFF_KT + 32 * B1_KT_PITCH)

typedef __attribute__((ext_vector_type(4))) short b1_s16x4;
__device__ __forceinline__ bf16x4 b1_tr_read(const unsigned char* p) {
  const b1_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) b1_s16x4*)p);
  return __builtin_bit_cast(bf16x4, v);
}
// swizzle of T's 8-byte chunks (16 per 128-byte row): a function of the row such that (i) 16 consecutive rows take 16
// different values (the ds_write_b64 of 16 lanes = 16 keys, same query chunk, hit 16 different bank pairs) and (ii) the
// 8 rows r .. r + 7 (r % 8 == 0) that one half-wave's transpose read touches -- 4 adjacent chunks each -- spread over
// the 2 x 4 (row parity, chunk group) combinations of the 64-bank read space.
__device__ __forceinline__ int b1_swz(int row) { return (((row >> 1) & 3) << 2) | (((row >> 3) & 1) << 1) | (row & 1); }

// hi plane only of two staged rows (DP1: the dO rows)
__device__ __forceinline__ void b1_put_rows_hi(__bf16* Th, int rp, int c0, float4 r0, float4 r1) {
  *reinterpret_cast<bf16x4*>(&Th[(2 * rp) * RP + c0]) = __builtin_convertvector(f32x4{r0.x, r0.y, r0.z, r0.w}, bf16x4);
  *reinterpret_cast<bf16x4*>(&Th[(2 * rp + 1) * RP + c0]) = __builtin_convertvector(f32x4{r1.x, r1.y, r1.z, r1.w}, bf16x4);
}

// DP1 (precision mode 3, "bf16 data gradients"): dP = dO V^T -- the data gradient of O = P V with respect to P -- as ONE
// bf16 product (dO, V rounded to bf16) like the convolutions' data gradients in that mode: 14 instead of 18
// MFMA-equivalents per tile, no lo plane of the staged dO rows.  S is a forward recomputation and stays split.
template <bool DROPOUT, bool DP1 = false>
__global__ __launch_bounds__(512, 2) void attn_bwd1_bx3_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ Dv, float* dQ,
    float* __restrict__ dK, float* __restrict__ dV, const uint32_t* __restrict__ MASK, int Ntok, int ld, int ldo, int ldg,
    float scale, float p_drop, int nheads) {
  extern __shared__ __attribute__((aligned(128))) unsigned char b1_smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int bh_ = blockIdx.x, H = nheads, h = bh_ % nheads, b = bh_ / nheads;
  const size_t base = (size_t)b * Ntok * ld + h * 32;
  const size_t baseo = (size_t)b * Ntok * ldo + h * 32;
  const size_t gbase = (size_t)b * Ntok * ldg + h * 32;
  const size_t sbase = (size_t)(b * H + h) * Ntok;
  const float inv_keep = DROPOUT ? 1.f / (1.f - (float)attn_drop_thr16(p_drop) / 65536.f) : 1.f;
  const int nq = Ntok / 64, nkc = Ntok / 256, NG = Ntok / 32, nit = nq * nkc;

  // ---- staging role: waves 0-3 stage Q (scaled to log2 units), waves 4-7 stage dO; thread = 2 rows x 4 columns
  const int ten = tid >> 8, t8 = tid & 255, rp = t8 >> 3, c0 = (t8 & 7) * 4;
  const float* const src = ten ? dO + baseo : Q + base;
  const int lds_ = ten ? ldo : ld;
  const float stg_scale = ten ? 1.f : scale * LOG2E;
  // ---- dQ role: wave = (16-query tile, 16-column half); lane = (query | column, k group)
  const int qt16 = wave & 3, dt = wave >> 2, la = lane & 15, kg = lane >> 4;
  const int trow0 = 16 * (kg >> 1) + 4 * (kg & 1) + (la >> 2);
  const int tchunk = 4 * qt16 + (la & 3);
  const int toff0 = trow0 * 128 + ((tchunk ^ b1_swz(trow0)) << 3);
  const int toff1 = (trow0 + 8) * 128 + ((tchunk ^ b1_swz(trow0 + 8)) << 3);
  const int ktoff = (16 * dt + la) * B1_KT_PITCH + 256 * (kg & 1) + 128 * (kg >> 1);
  float* const dqp = dQ + gbase + (size_t)(16 * qt16 + la) * ldg + 16 * dt + 4 * kg;
  // ---- T write role: row = this lane's key in the chunk, chunk = sub * 8 + 2 g + lh
  const int krow = wave * 32 + li;
  const int twoff = krow * 128 + (((lh ^ b1_swz(krow)) & 15) << 3);
  // position of this lane's key inside a K^T row: (k step s, k group, slot e) <-> key 32 s + 16 (kg >> 1) + 8 (e >> 2) +
  // 4 (kg & 1) + (e & 3), stored at byte 256 (kg & 1) + 128 (kg >> 1) + 16 s + 2 e
  const int kpos = 256 * ((li >> 2) & 1) + 128 * ((li >> 4) & 1) + 16 * wave + 2 * (4 * ((li >> 3) & 1) + (li & 3));

  float4 r0, r1;
  float lreg = 0.f;
  const float* const lsd_src = (ten ? Dv : LSE) + sbase + lane;
  uint32_t mreg0 = 0u, mreg1 = 0u;
#define B1_LOAD(it_)                                                                               \
  do {                                                                                             \
    const int kc_ = (it_) / nq, qt_ = (it_) - kc_ * nq;                                            \
    const float* p_ = src + (size_t)(qt_ * 64 + 2 * rp) * lds_ + c0;                               \
    r0 = *reinterpret_cast<const float4*>(p_);                                                     \
    r1 = *reinterpret_cast<const float4*>(p_ + lds_);                                              \
    lreg = lsd_src[qt_ * 64];          /* every wave requests (no branch); waves 0 / 4 store LSE / D */ \
    if (DROPOUT) {                                                                                 \
      const uint32_t* mk_ = MASK + ((size_t)bh_ * NG * NG + (kc_ * 8 + wave)) * 32 + mask_slot(li); \
      mreg0 = mk_[(size_t)(qt_ * 2) * NG * 32];                                                    \
      mreg1 = mk_[(size_t)(qt_ * 2 + 1) * NG * 32];                                                \
    }                                                                                              \
  } while (0)
#define B1_STORE(buf_)                                                                             \
  do {                                                                                             \
    __bf16* st_ = reinterpret_cast<__bf16*>(b1_smem + (buf_) * B1_STAGE);                          \
    __bf16* rh_ = st_ + ten * (2 * 64 * RP);                                                       \
    __bf16* th_ = st_ + 4 * 64 * RP + ten * (32 * TP);                                             \
    const float4 a_ = scale4(r0, stg_scale), b_ = scale4(r1, stg_scale);                           \
    if (DP1 && ten) b1_put_rows_hi(rh_, rp, c0, a_, b_);                                           \
    else put_rows(rh_, rh_ + 64 * RP, rp, c0, a_, b_);                                             \
    put_cols_hi(th_, rp, c0, a_, b_);                                                              \
    float* ls_ = reinterpret_cast<float*>(b1_smem + B1_OFF_LS);                                    \
    if (t8 < 64) ls_[128 * ten + (buf_) * 64 + t8] = ten ? lreg : -lreg * LOG2E;                   \
  } while (0)

  int boff[16];                                    // keep-bit offsets of the 16 accumulator registers, opaque SGPRs
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    boff[r] = (r & 3) + 8 * (r >> 2);
    asm volatile("" : "+s"(boff[r]));
  }
  bf16x8 kh[2], kl[2], vh[2], vl[2];
  f32x16 dkacc, dvacc;
  float* const lsds = reinterpret_cast<float*>(b1_smem + B1_OFF_LS);

  // ---- dQ^T[d][q] += K^T[d][key] dS^T[key][q] over the chunk's 256 keys for the tile of iteration `jt` (T buffer jt & 1):
  // 8 x v_mfma_f32_16x16x32_bf16 per wave, two accumulator chains.  Split in request / product halves so that the tile
  // code below can place them where the LDS latency and the matrix-pipe time are covered.
  struct DqFrag { bf16x8 a[4], b[4]; };
  using I0 = std::integral_constant<int, 0>; using I8 = std::integral_constant<int, 8>; using I16 = std::integral_constant<int, 16>;
  using I12 = std::integral_constant<int, 12>;
  auto dq_request = [&](int jt, int half, DqFrag& f) {
    const unsigned char* Tr = b1_smem + B1_OFF_T + (jt & 1) * B1_T_BYTES;
    const unsigned char* Kt = b1_smem + B1_OFF_KT + ktoff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s8 = 4 * half + j;
      f.a[j] = *reinterpret_cast<const bf16x8*>(Kt + 16 * s8);
      f.b[j] = cat44(b1_tr_read(Tr + toff0 + 4096 * s8), b1_tr_read(Tr + toff1 + 4096 * s8));
    }
  };
  auto dq_product = [&](const DqFrag& f, f32x4& acc0, f32x4& acc1) {
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[0], f.b[0], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[1], f.b[1], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[2], f.b[2], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[3], f.b[3], acc1, 0, 0, 0);
  };
  auto dq_store = [&](int jt, const f32x4& acc0, const f32x4& acc1, float4 prev, bool have_prev) {
    float* const row = dqp + (size_t)((jt % nq) * 64) * ldg;
    if (!have_prev) prev = make_float4(0.f, 0.f, 0.f, 0.f);      // (a select, not a branch)
    *reinterpret_cast<float4*>(row) =
        make_float4(fmaf(acc0[0] + acc1[0], scale, prev.x), fmaf(acc0[1] + acc1[1], scale, prev.y),
                    fmaf(acc0[2] + acc1[2], scale, prev.z), fmaf(acc0[3] + acc1[3], scale, prev.w));
  };
  auto dq_phase = [&](int jt, float4 prev, bool have_prev) {       // the whole product in one go (chunk boundaries, tail)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    DqFrag f;
    dq_request(jt, 0, f);
    dq_product(f, acc0, acc1);
    dq_request(jt, 1, f);
    dq_product(f, acc0, acc1);
    dq_store(jt, acc0, acc1, prev, have_prev);
  };

  // ---- pieces of one 32-query sub-tile (S, dP, dS, dV, dK; dS -> T[p]) of the staged tile in buffer p
  struct RowFrag { bf16x8 qh[2], ql[2], gh[2], gl[2]; };
  auto st_ptr = [&](int p) { return reinterpret_cast<const __bf16*>(b1_smem + p * B1_STAGE); };
  // s starts at -LSE of its query row (register r <-> query key_of_b(r, lh)): the MFMAs deliver s - lse for free
  auto req_lse = [&](int p, int sub, f32x16& s) {
    const float* Ls = lsds + p * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 l4 = *reinterpret_cast<const float4*>(&Ls[sub * 32 + 8 * g + 4 * lh]);
      s[4 * g] = l4.x; s[4 * g + 1] = l4.y; s[4 * g + 2] = l4.z; s[4 * g + 3] = l4.w;
    }
  };
  auto req_d = [&](int p, int sub, float (&dd)[16]) {
    const float* Ds = lsds + 128 + p * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 d4 = *reinterpret_cast<const float4*>(&Ds[sub * 32 + 8 * g + 4 * lh]);
      dd[4 * g] = d4.x; dd[4 * g + 1] = d4.y; dd[4 * g + 2] = d4.z; dd[4 * g + 3] = d4.w;
    }
  };
  auto req_rows = [&](int p, int sub, RowFrag& f) {
    const __bf16* st = st_ptr(p);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int off = (sub * 32 + li) * RP + 16 * m + 8 * lh;
      f.qh[m] = *reinterpret_cast<const bf16x8*>(&st[off]);
      f.ql[m] = *reinterpret_cast<const bf16x8*>(&st[64 * RP + off]);
      f.gh[m] = *reinterpret_cast<const bf16x8*>(&st[2 * 64 * RP + off]);
      if constexpr (!DP1) f.gl[m] = *reinterpret_cast<const bf16x8*>(&st[3 * 64 * RP + off]);
    }
  };
  struct ColFrag { bf16x8 qt[2], gt[2]; };
  auto req_cols = [&](int p, int sub, ColFrag& f) {
    const __bf16 *Qth = st_ptr(p) + 4 * 64 * RP, *Gth = Qth + 32 * TP;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int qc = sub * 32 + 16 * m + 4 * lh;
      f.gt[m] = cat44(*reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc]), *reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc + 8]));
      f.qt[m] = cat44(*reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc]), *reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc + 8]));
    }
  };
  // the 12 score / dP products, alternating between the two accumulators (consecutive MFMAs are independent: other
  // instructions may sit between them at no cost)
#define B1_MFMA_PAIR(s_, dp_, f_, m_, A_, B_)                                                                   \
  do {                                                                                                          \
    s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_ ? f_.ql[m_] : f_.qh[m_], B_ ? kl[m_] : kh[m_], s_, 0, 0, 0);   \
    dp_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_ ? f_.gl[m_] : f_.gh[m_], B_ ? vl[m_] : vh[m_], dp_, 0, 0, 0); \
  } while (0)
  auto scores = [&](f32x16& s, f32x16& dp, const RowFrag& f) {
#ifndef B1_ABL_MFMA12
    if constexpr (DP1) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.qh[m], kh[m], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.gh[m], vh[m], dp, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.qh[m], kl[m], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ql[m], kh[m], s, 0, 0, 0);
      }
    } else {
      B1_MFMA_PAIR(s, dp, f, 0, 0, 0); B1_MFMA_PAIR(s, dp, f, 0, 0, 1); B1_MFMA_PAIR(s, dp, f, 0, 1, 0);
      B1_MFMA_PAIR(s, dp, f, 1, 0, 0); B1_MFMA_PAIR(s, dp, f, 1, 0, 1); B1_MFMA_PAIR(s, dp, f, 1, 1, 0);
    }
#else
    s[0] += (float)f.qh[0][0] + (float)f.ql[1][1]; dp[0] += (float)f.gh[0][0] + (float)f.gl[1][0];
#endif
  };
  // dS = P (M dP' - D) = (M P) dP' - P D with M the keep mask: one AND instead of two, the rest an fma.  On return s
  // holds M P (the dV operand), dp holds dS.  Registers [lo, hi).
  auto softmax_grad = [&](f32x16& s, f32x16& dp, const float (&dd)[16], uint32_t mcur, auto lo, auto hi) {
#ifndef B1_ABL_VALU
#pragma unroll
    for (int r = decltype(lo)::value; r < decltype(hi)::value; ++r) {
      const float pr = __builtin_amdgcn_exp2f(s[r]);
      float pd = pr;
      if (DROPOUT) {
        const int mk = bit_sext(mcur, boff[r]);                     // query bit of this lane's key word (SGPR offset: 2 VALU)
        pd = __int_as_float(__float_as_int(pr) & mk);               // 1/(1-p) folded into the dV store
      }
      s[r] = pd;
      dp[r] = fmaf(pd, dp[r], -pr * dd[r]);
    }
#else
    if (decltype(lo)::value == 0) dp[0] += dd[0] + dd[5] + dd[10] + dd[15];
#endif
  };
  auto grads = [&](int p, int sub, const f32x16& s, const f32x16& dp, const ColFrag& f) {
    unsigned char* const Tb = b1_smem + B1_OFF_T + p * B1_T_BYTES;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 ph, sh;
      hi_regs(s, m, ph);
      hi_regs(dp, m, sh);
      MFMA1(dvacc, f.gt[m], ph);
      MFMA1(dkacc, f.qt[m], sh);
      // registers 8 m .. 8 m + 3 / + 4 .. + 7 = queries 8 g + 4 lh + (0 .. 3) of the sub-tile, g = 2 m / 2 m + 1
      const uint4 w = __builtin_bit_cast(uint4, sh);
#ifndef B1_ABL_T
      *reinterpret_cast<uint2*>(Tb + (twoff ^ ((sub * 8 + 4 * m) << 3))) = make_uint2(w.x, w.y);
      *reinterpret_cast<uint2*>(Tb + (twoff ^ ((sub * 8 + 4 * m + 2) << 3))) = make_uint2(w.z, w.w);
#endif
    }
  };

#ifdef B1_PRIO
  if (__builtin_amdgcn_readfirstlane(ten) != 0) __builtin_amdgcn_s_setprio(1);
#endif
  B1_LOAD(0);
  B1_STORE(0);
  for (int kc = 0; kc < nkc; ++kc) {
    const int key = kc * 256 + krow;
    // ---- chunk begin: the dQ product of the previous chunk's last tile still needs the old K^T
    if (kc > 0) {
      const int jt = kc * nq - 1;
      float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kc > 1) prev = *reinterpret_cast<const float4*>(dqp + (size_t)((nq - 1) * 64) * ldg);
      dq_phase(jt, prev, kc > 1);
    }
    // this wave's 32 keys (K / V fragments, zeroed accumulators), K^T of the chunk for the dQ product
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      row_frag(K + base + (size_t)key * ld + 16 * m + 8 * lh, 1.f, kh[m], kl[m]);
      // V carries the dropout scale 1/(1-p): dP' = dO (V/(1-p))^T is all the dS formula below needs of it
      row_frag(V + base + (size_t)key * ld + 16 * m + 8 * lh, inv_keep, vh[m], vl[m]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkacc[r] = 0.f; dvacc[r] = 0.f; }
    __syncthreads();                                   // every wave is done with the previous chunk's K^T
    {
      __bf16* kt_ = reinterpret_cast<__bf16*>(b1_smem + B1_OFF_KT + kpos);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) kt_[(16 * m + 8 * lh + e) * (B1_KT_PITCH / 2)] = kh[m][e];
    }
    // one query tile.  WITH_DQ: the dQ product of the PREVIOUS tile (its T buffer is complete since the last barrier) shares
    // this stretch of straight-line code with the S / dP products of the current tile, so that the LDS reads and the
    // MFMAs of the two interleave; the first tile of a chunk has none pending (done at the chunk boundary above)
    auto tile = [&](auto with_dq, int qt) {
      const int it = kc * nq + qt, p = it & 1;
      // (consume the keep words requested one iteration ago BEFORE this iteration's requests go out: vmcnt retires in
      // order, a wait placed behind the new requests would wait for them too)
      uint32_t mcur0 = mreg0 >> (4 * lh), mcur1 = mreg1 >> (4 * lh);
      asm volatile("" : "+v"(mcur0), "+v"(mcur1));
      __builtin_amdgcn_sched_barrier(0);
      // tile it + 1 (clamped: the very last iteration reloads its own tile into the idle buffer) is requested a whole
      // iteration before its LDS store; the previous tile's dQ rows from one chunk ago likewise
      const int nxt = it + 1 < nit ? it + 1 : it;
      // (unconditional load, FIRST of the iteration's requests so that waiting for it does not wait for the tile loads
      // behind it: before the first chunk has written them the rows hold whatever the caller left there, and dq_phase
      // ignores the value)
      float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (decltype(with_dq)::value) prev = *reinterpret_cast<const float4*>(dqp + (size_t)((qt - 1) * 64) * ldg);
#ifndef B1_ABL_STAGE
      B1_LOAD(nxt);
#endif
      __builtin_amdgcn_sched_barrier(0);         // keep the requests up here (the scheduler sinks them to their uses)
      constexpr bool WDQ = decltype(with_dq)::value;
      // ---- a hand-ordered software pipeline (sched_barrier between the stages keeps hipcc from re-serialising it):
      // every LDS request is issued a stage before its consumer, and the VALU work of sub-tile 0 (exp2, keep bit, dS)
      // sits between the score MFMAs of sub-tile 1
      f32x16 s0, dp0, s1, dp1;
      float dd0[16], dd1[16];
      RowFrag rf;
      ColFrag cf;
      DqFrag qf;
      f32x4 qa0, qa1;
      // stage A: requests of sub-tile 0
      req_lse(p, 0, s0);
      req_rows(p, 0, rf);
      req_d(p, 0, dd0);
#pragma unroll
      for (int r = 0; r < 16; ++r) dp0[r] = 0.f;
      __builtin_amdgcn_sched_barrier(0);
      // stage B: scores of sub-tile 0; requests: sub-tile 1's rows
      scores(s0, dp0, rf);
      req_lse(p, 1, s1);
      req_rows(p, 1, rf);
      __builtin_amdgcn_sched_barrier(0);
      // stage C: VALU of sub-tile 0 between the score MFMAs of sub-tile 1
#pragma unroll
      for (int r = 0; r < 16; ++r) dp1[r] = 0.f;
      scores(s1, dp1, rf);
      softmax_grad(s0, dp0, dd0, mcur0, I0{}, I12{});
#if !defined(B1_ABL_VALU) && !defined(B1_ABL_MFMA12) && !defined(B1_NO_GROUPS)
      // MFMA = 0x8, VALU = 0x2, TRANS = 0x400: one matrix instruction, then one score register's VALU work (12 of 16 registers here)
#define B1_GROUP                                                   \
  __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                 \
  __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x2, DROPOUT ? 4 : 2, 0);
#define B1_GROUP_R                                                 \
  __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);               \
  __builtin_amdgcn_sched_group_barrier(0x2, DROPOUT ? 4 : 2, 0);
      if constexpr (DP1) {      // 8 matrix instructions over the same 12 registers' VALU work
        B1_GROUP B1_GROUP B1_GROUP_R B1_GROUP B1_GROUP B1_GROUP_R B1_GROUP B1_GROUP B1_GROUP_R B1_GROUP B1_GROUP B1_GROUP_R
      } else {
        B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP B1_GROUP
      }
#undef B1_GROUP_R
#undef B1_GROUP
#endif
      __builtin_amdgcn_sched_barrier(0);
      // (the column fragments are requested only now -- sub-tile 1's row fragments have just died -- and the last quarter
      // of the VALU work covers their latency)
      req_cols(p, 0, cf);
      req_d(p, 1, dd1);
      __builtin_amdgcn_sched_barrier(0);
      softmax_grad(s0, dp0, dd0, mcur0, I12{}, I16{});
      __builtin_amdgcn_sched_barrier(0);
      // stage D: dV / dK of sub-tile 0, dS -> T; requests: sub-tile 1's columns, first half of the pending dQ product
      grads(p, 0, s0, dp0, cf);
      req_cols(p, 1, cf);
#ifndef B1_ABL_DQ
      if constexpr (WDQ) dq_request(it - 1, 0, qf);
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) { qa0[r] = 0.f; qa1[r] = 0.f; }
      __builtin_amdgcn_sched_barrier(0);
      // stage E: VALU of sub-tile 1 -- the one stretch without score MFMAs of its own -- around the pending dQ product of
      // the PREVIOUS tile (its T buffer is complete since the last barrier)
      softmax_grad(s1, dp1, dd1, mcur1, I0{}, I8{});
#ifndef B1_ABL_DQ
      if constexpr (WDQ) {
        dq_product(qf, qa0, qa1);
        __builtin_amdgcn_sched_barrier(0);
        dq_request(it - 1, 1, qf);
      }
#endif
      softmax_grad(s1, dp1, dd1, mcur1, I8{}, I16{});
#ifndef B1_ABL_DQ
      if constexpr (WDQ) {
        dq_product(qf, qa0, qa1);
        dq_store(it - 1, qa0, qa1, prev, kc > 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      grads(p, 1, s1, dp1, cf);
#ifndef B1_ABL_STAGE
      B1_STORE(p ^ 1);
#endif
#ifndef B1_ABL_BAR
      __syncthreads();
#endif
    };
    tile(std::false_type{}, 0);
    for (int qt = 1; qt < nq; ++qt) tile(std::true_type{}, qt);
    {
      float* dkrow = dK + gbase + (size_t)key * ldg;
      float* dvrow = dV + gbase + (size_t)key * ldg;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dkrow + 8 * g + 4 * lh) =
            make_float4(dkacc[4 * g] * LN2, dkacc[4 * g + 1] * LN2, dkacc[4 * g + 2] * LN2, dkacc[4 * g + 3] * LN2);
        *reinterpret_cast<float4*>(dvrow + 8 * g + 4 * lh) = make_float4(
            dvacc[4 * g] * inv_keep, dvacc[4 * g + 1] * inv_keep, dvacc[4 * g + 2] * inv_keep, dvacc[4 * g + 3] * inv_keep);
      }
    }
  }
  {   // the last tile's dQ product
    float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nkc > 1) prev = *reinterpret_cast<const float4*>(dqp + (size_t)((nq - 1) * 64) * ldg);
    dq_phase(nit - 1, prev, nkc > 1);
  }
#undef B1_LOAD
#undef B1_STORE
}

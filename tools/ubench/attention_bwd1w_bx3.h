// Single-pass attention backward, ONE wave per SIMD (round 4, experiment on top of attention_bwd1_bx3.h; same LDS image,
// same arithmetic, same results).  Four waves per block, each owning TWO key tiles (64 keys) of the 256-key chunk and the
// whole 512-entry register file: the four (sub-tile, key-tile) units of a query tile form one software pipeline inside
// the wave -- the score MFMAs of unit u + 1 sit between the exp2 / keep-bit / dS instructions of unit u by construction
// instead of by the arbitration between two waves that run the same phase at the same time -- and every Q / dO fragment
// read from LDS feeds two key tiles.
#define W1_KT 2

template <bool DROPOUT>
__global__ __launch_bounds__(256, 1) void attn_bwd1w_bx3_kernel(
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

  // ---- staging: every thread stages 2 rows x 4 columns of BOTH tensors
  const int rp = tid >> 3, c0 = (tid & 7) * 4;
  // ---- dQ role: wave = 16-query tile, both 16-column halves; lane = (query | column, k group)
  const int la = lane & 15, kg = lane >> 4;
  const int trow0 = 16 * (kg >> 1) + 4 * (kg & 1) + (la >> 2);
  const int tchunk = 4 * wave + (la & 3);
  const int toff0 = trow0 * 128 + ((tchunk ^ b1_swz(trow0)) << 3);
  const int toff1 = (trow0 + 8) * 128 + ((tchunk ^ b1_swz(trow0 + 8)) << 3);
  const int ktoff = la * B1_KT_PITCH + 256 * (kg & 1) + 128 * (kg >> 1);          // + 16 rows for the second column half
  float* const dqp = dQ + gbase + (size_t)(16 * wave + la) * ldg + 4 * kg;        // + 16 for the second column half
  // ---- per key tile: T row, K^T position
  int twoff[W1_KT], kpos[W1_KT];
#pragma unroll
  for (int kt = 0; kt < W1_KT; ++kt) {
    const int krow = wave * 64 + kt * 32 + li;
    twoff[kt] = krow * 128 + (((lh ^ b1_swz(krow)) & 15) << 3);
    kpos[kt] = 256 * ((li >> 2) & 1) + 128 * ((li >> 4) & 1) + 16 * (2 * wave + kt) + 2 * (4 * ((li >> 3) & 1) + (li & 3));
  }

  float4 rq0, rq1, rg0, rg1;
  float lreg = 0.f;
  const float* const lsd_src = ((wave & 1) ? Dv : LSE) + sbase + lane;       // wave 0 stages -LSE, wave 1 stages D
  uint32_t mreg[W1_KT][2] = {{0u, 0u}, {0u, 0u}};
#define W1_LOAD_TILE(it_)                                                                          \
  do {                                                                                             \
    const int kc_ = (it_) / nq, qt_ = (it_) - kc_ * nq;                                            \
    const float* pq_ = Q + base + (size_t)(qt_ * 64 + 2 * rp) * ld + c0;                           \
    const float* pg_ = dO + baseo + (size_t)(qt_ * 64 + 2 * rp) * ldo + c0;                        \
    rq0 = *reinterpret_cast<const float4*>(pq_);                                                   \
    rq1 = *reinterpret_cast<const float4*>(pq_ + ld);                                              \
    rg0 = *reinterpret_cast<const float4*>(pg_);                                                   \
    rg1 = *reinterpret_cast<const float4*>(pg_ + ldo);                                             \
    lreg = lsd_src[qt_ * 64];                                                                      \
    if (DROPOUT) {                                                                                 \
      _Pragma("unroll") for (int kt_ = 0; kt_ < W1_KT; ++kt_) {                                    \
        const uint32_t* mk_ = MASK + ((size_t)bh_ * NG * NG + (kc_ * 8 + 2 * wave + kt_)) * 32 + mask_slot(li); \
        mreg[kt_][0] = mk_[(size_t)(qt_ * 2) * NG * 32];                                           \
        mreg[kt_][1] = mk_[(size_t)(qt_ * 2 + 1) * NG * 32];                                       \
      }                                                                                            \
    }                                                                                              \
  } while (0)
#define W1_STORE(buf_)                                                                             \
  do {                                                                                             \
    __bf16* st_ = reinterpret_cast<__bf16*>(b1_smem + (buf_) * B1_STAGE);                          \
    const float4 a_ = scale4(rq0, scale * LOG2E), b_ = scale4(rq1, scale * LOG2E);                 \
    put_rows(st_, st_ + 64 * RP, rp, c0, a_, b_);                                                  \
    put_cols_hi(st_ + 4 * 64 * RP, rp, c0, a_, b_);                                                \
    put_rows(st_ + 2 * 64 * RP, st_ + 3 * 64 * RP, rp, c0, rg0, rg1);                              \
    put_cols_hi(st_ + 4 * 64 * RP + 32 * TP, rp, c0, rg0, rg1);                                    \
    float* ls_ = reinterpret_cast<float*>(b1_smem + B1_OFF_LS);                                    \
    if (wave < 2) ls_[128 * wave + (buf_) * 64 + lane] = wave ? lreg : -lreg * LOG2E;              \
  } while (0)

  int boff[16];                                    // keep-bit offsets of the 16 accumulator registers, opaque SGPRs
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    boff[r] = (r & 3) + 8 * (r >> 2);
    asm volatile("" : "+s"(boff[r]));
  }
  bf16x8 kh[W1_KT][2], kl[W1_KT][2], vh[W1_KT][2], vl[W1_KT][2];
  f32x16 dkacc[W1_KT], dvacc[W1_KT];
  float* const lsds = reinterpret_cast<float*>(b1_smem + B1_OFF_LS);
  using I0 = std::integral_constant<int, 0>; using I8 = std::integral_constant<int, 8>;
  using I12 = std::integral_constant<int, 12>; using I16 = std::integral_constant<int, 16>;

  // ---- dQ^T[d][q] += K^T[d][key] dS^T[key][q]: this wave's 16 queries x both 16-column halves over the chunk's 256 keys
  struct DqFragW { bf16x8 a[2][4], b[4]; };
  auto dq_request = [&](int jt, int half, DqFragW& f) {
    const unsigned char* Tr = b1_smem + B1_OFF_T + (jt & 1) * B1_T_BYTES;
    const unsigned char* Kt = b1_smem + B1_OFF_KT + ktoff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s8 = 4 * half + j;
      f.a[0][j] = *reinterpret_cast<const bf16x8*>(Kt + 16 * s8);
      f.a[1][j] = *reinterpret_cast<const bf16x8*>(Kt + 16 * B1_KT_PITCH + 16 * s8);
      f.b[j] = cat44(b1_tr_read(Tr + toff0 + 4096 * s8), b1_tr_read(Tr + toff1 + 4096 * s8));
    }
  };
  auto dq_product = [&](const DqFragW& f, f32x4 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        acc[dt][j & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[dt][j], f.b[j], acc[dt][j & 1], 0, 0, 0);
  };
  auto dq_store = [&](int jt, const f32x4 (&acc)[2][2], const float4 (&prev)[2], bool have_prev) {
    float* const row = dqp + (size_t)((jt % nq) * 64) * ldg;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float4 pv = prev[dt];
      if (!have_prev) pv = make_float4(0.f, 0.f, 0.f, 0.f);      // (a select, not a branch)
      *reinterpret_cast<float4*>(row + 16 * dt) = make_float4(
          fmaf(acc[dt][0][0] + acc[dt][1][0], scale, pv.x), fmaf(acc[dt][0][1] + acc[dt][1][1], scale, pv.y),
          fmaf(acc[dt][0][2] + acc[dt][1][2], scale, pv.z), fmaf(acc[dt][0][3] + acc[dt][1][3], scale, pv.w));
    }
  };
  auto dq_zero = [&](f32x4 (&acc)[2][2]) {
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[dt][u][r] = 0.f;
  };
  auto dq_phase = [&](int jt, const float4 (&prev)[2], bool have_prev) {       // the whole product in one go
    f32x4 acc[2][2];
    dq_zero(acc);
    DqFragW f;
    dq_request(jt, 0, f);
    dq_product(f, acc);
    dq_request(jt, 1, f);
    dq_product(f, acc);
    dq_store(jt, acc, prev, have_prev);
  };
  auto dq_prev = [&](int qrow, float4 (&prev)[2]) {
    prev[0] = *reinterpret_cast<const float4*>(dqp + (size_t)qrow * ldg);
    prev[1] = *reinterpret_cast<const float4*>(dqp + (size_t)qrow * ldg + 16);
  };

  // ---- pieces of one 32-query sub-tile of the staged tile in buffer p (Q / dO fragments shared by both key tiles)
  struct RowFrag { bf16x8 qh[2], ql[2], gh[2], gl[2]; };
  struct ColFrag { bf16x8 qt[2], gt[2]; };
  auto st_ptr = [&](int p) { return reinterpret_cast<const __bf16*>(b1_smem + p * B1_STAGE); };
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
      f.gl[m] = *reinterpret_cast<const bf16x8*>(&st[3 * 64 * RP + off]);
    }
  };
  auto req_cols = [&](int p, int sub, ColFrag& f) {
    const __bf16 *Qth = st_ptr(p) + 4 * 64 * RP, *Gth = Qth + 32 * TP;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int qc = sub * 32 + 16 * m + 4 * lh;
      f.gt[m] = cat44(*reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc]), *reinterpret_cast<const bf16x4*>(&Gth[li * TP + qc + 8]));
      f.qt[m] = cat44(*reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc]), *reinterpret_cast<const bf16x4*>(&Qth[li * TP + qc + 8]));
    }
  };
#define W1_MFMA_PAIR(s_, dp_, f_, kt_, m_, A_, B_)                                                              \
  do {                                                                                                          \
    s_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_ ? f_.ql[m_] : f_.qh[m_], B_ ? kl[kt_][m_] : kh[kt_][m_], s_, 0, 0, 0);   \
    dp_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_ ? f_.gl[m_] : f_.gh[m_], B_ ? vl[kt_][m_] : vh[kt_][m_], dp_, 0, 0, 0); \
  } while (0)
  auto scores = [&](auto ktc, f32x16& s, f32x16& dp, const RowFrag& f) {
    constexpr int kt = decltype(ktc)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
    W1_MFMA_PAIR(s, dp, f, kt, 0, 0, 0); W1_MFMA_PAIR(s, dp, f, kt, 0, 0, 1); W1_MFMA_PAIR(s, dp, f, kt, 0, 1, 0);
    W1_MFMA_PAIR(s, dp, f, kt, 1, 0, 0); W1_MFMA_PAIR(s, dp, f, kt, 1, 0, 1); W1_MFMA_PAIR(s, dp, f, kt, 1, 1, 0);
  };
  auto softmax_grad = [&](f32x16& s, f32x16& dp, const float (&dd)[16], uint32_t mcur, auto lo, auto hi) {
#pragma unroll
    for (int r = decltype(lo)::value; r < decltype(hi)::value; ++r) {
      const float pr = __builtin_amdgcn_exp2f(s[r]);
      float pd = pr;
      if (DROPOUT) {
        const int mk = bit_sext(mcur, boff[r]);
        pd = __int_as_float(__float_as_int(pr) & mk);
      }
      s[r] = pd;
      dp[r] = fmaf(pd, dp[r], -pr * dd[r]);
    }
  };
  auto grads = [&](int p, int sub, auto ktc, const f32x16& s, const f32x16& dp, const ColFrag& f) {
    constexpr int kt = decltype(ktc)::value;
    unsigned char* const Tb = b1_smem + B1_OFF_T + p * B1_T_BYTES;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      bf16x8 ph, sh;
      hi_regs(s, m, ph);
      hi_regs(dp, m, sh);
      MFMA1(dvacc[kt], f.gt[m], ph);
      MFMA1(dkacc[kt], f.qt[m], sh);
      const uint4 w = __builtin_bit_cast(uint4, sh);
      *reinterpret_cast<uint2*>(Tb + (twoff[kt] ^ ((sub * 8 + 4 * m) << 3))) = make_uint2(w.x, w.y);
      *reinterpret_cast<uint2*>(Tb + (twoff[kt] ^ ((sub * 8 + 4 * m + 2) << 3))) = make_uint2(w.z, w.w);
    }
  };
  // scores of one unit between the VALU work of another: MFMA = 0x8, VALU = 0x2, TRANS = 0x400 (one matrix instruction,
  // then one score register's exp2 / keep bit / dS)
#define W1_INTERLEAVE_12                                           \
  do {                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) {            \
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);             \
      __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);           \
      __builtin_amdgcn_sched_group_barrier(0x2, DROPOUT ? 4 : 2, 0); \
    }                                                              \
  } while (0)
  using KT0 = std::integral_constant<int, 0>;
  using KT1 = std::integral_constant<int, 1>;

  W1_LOAD_TILE(0);
  W1_STORE(0);
  for (int kc = 0; kc < nkc; ++kc) {
    // ---- chunk begin: the dQ product of the previous chunk's last tile still needs the old K^T
    if (kc > 0) {
      float4 prev[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      if (kc > 1) dq_prev((nq - 1) * 64, prev);
      dq_phase(kc * nq - 1, prev, kc > 1);
    }
#pragma unroll
    for (int kt = 0; kt < W1_KT; ++kt) {
      const int key = kc * 256 + wave * 64 + kt * 32 + li;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        row_frag(K + base + (size_t)key * ld + 16 * m + 8 * lh, 1.f, kh[kt][m], kl[kt][m]);
        row_frag(V + base + (size_t)key * ld + 16 * m + 8 * lh, inv_keep, vh[kt][m], vl[kt][m]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkacc[kt][r] = 0.f; dvacc[kt][r] = 0.f; }
    }
    __syncthreads();                                   // every wave is done with the previous chunk's K^T
#pragma unroll
    for (int kt = 0; kt < W1_KT; ++kt) {
      __bf16* kt_ = reinterpret_cast<__bf16*>(b1_smem + B1_OFF_KT + kpos[kt]);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 8; ++e) kt_[(16 * m + 8 * lh + e) * (B1_KT_PITCH / 2)] = kh[kt][m][e];
    }
    auto tile = [&](auto with_dq, int qt) {
      constexpr bool WDQ = decltype(with_dq)::value;
      const int it = kc * nq + qt, p = it & 1;
      uint32_t mc[W1_KT][2];
#pragma unroll
      for (int kt = 0; kt < W1_KT; ++kt) {
        mc[kt][0] = mreg[kt][0] >> (4 * lh);
        mc[kt][1] = mreg[kt][1] >> (4 * lh);
        asm volatile("" : "+v"(mc[kt][0]), "+v"(mc[kt][1]));
      }
      __builtin_amdgcn_sched_barrier(0);
      const int nxt = it + 1 < nit ? it + 1 : it;
      float4 prev[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      if constexpr (WDQ) dq_prev((qt - 1) * 64, prev);
      W1_LOAD_TILE(nxt);
      __builtin_amdgcn_sched_barrier(0);
      f32x16 sa0, da0, sa1, da1, sb0, db0, sb1, db1;       // (sub-tile a / b) x (key tile 0 / 1)
      float dd0[16], dd1[16];
      RowFrag rf;
      ColFrag cf;
      DqFragW qf;
      f32x4 qa[2][2];
      // A: requests of sub-tile 0
      req_lse(p, 0, sa0);
      req_lse(p, 0, sa1);
      req_rows(p, 0, rf);
      req_d(p, 0, dd0);
      __builtin_amdgcn_sched_barrier(0);
      // B: scores (0, 0)
      scores(KT0{}, sa0, da0, rf);
      __builtin_amdgcn_sched_barrier(0);
      // C: scores (0, 1) between the VALU of (0, 0)
      scores(KT1{}, sa1, da1, rf);
      softmax_grad(sa0, da0, dd0, mc[0][0], I0{}, I12{});
      W1_INTERLEAVE_12;
      __builtin_amdgcn_sched_barrier(0);
      req_cols(p, 0, cf);
      req_lse(p, 1, sb0);
      req_lse(p, 1, sb1);
      req_rows(p, 1, rf);
      __builtin_amdgcn_sched_barrier(0);
      softmax_grad(sa0, da0, dd0, mc[0][0], I12{}, I16{});
      __builtin_amdgcn_sched_barrier(0);
      // D: gradients of (0, 0); E: scores (1, 0) between the VALU of (0, 1)
      grads(p, 0, KT0{}, sa0, da0, cf);
      __builtin_amdgcn_sched_barrier(0);
      scores(KT0{}, sb0, db0, rf);
      softmax_grad(sa1, da1, dd0, mc[1][0], I0{}, I12{});
      W1_INTERLEAVE_12;
      __builtin_amdgcn_sched_barrier(0);
      req_d(p, 1, dd1);
      softmax_grad(sa1, da1, dd0, mc[1][0], I12{}, I16{});
      __builtin_amdgcn_sched_barrier(0);
      // F: gradients of (0, 1); G: scores (1, 1) between the VALU of (1, 0)
      grads(p, 0, KT1{}, sa1, da1, cf);
      __builtin_amdgcn_sched_barrier(0);
      scores(KT1{}, sb1, db1, rf);
      softmax_grad(sb0, db0, dd1, mc[0][1], I0{}, I12{});
      W1_INTERLEAVE_12;
      __builtin_amdgcn_sched_barrier(0);
      req_cols(p, 1, cf);
      if constexpr (WDQ) dq_request(it - 1, 0, qf);
      dq_zero(qa);
      __builtin_amdgcn_sched_barrier(0);
      softmax_grad(sb0, db0, dd1, mc[0][1], I12{}, I16{});
      __builtin_amdgcn_sched_barrier(0);
      // H: gradients of (1, 0); I: VALU of (1, 1) around the pending dQ product of the PREVIOUS tile
      grads(p, 1, KT0{}, sb0, db0, cf);
      __builtin_amdgcn_sched_barrier(0);
      softmax_grad(sb1, db1, dd1, mc[1][1], I0{}, I8{});
      if constexpr (WDQ) {
        dq_product(qf, qa);
        __builtin_amdgcn_sched_barrier(0);
        dq_request(it - 1, 1, qf);
      }
      softmax_grad(sb1, db1, dd1, mc[1][1], I8{}, I16{});
      if constexpr (WDQ) {
        dq_product(qf, qa);
        dq_store(it - 1, qa, prev, kc > 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // J: gradients of (1, 1); next tile into the other staging buffer
      grads(p, 1, KT1{}, sb1, db1, cf);
      W1_STORE(p ^ 1);
      __syncthreads();
    };
    tile(std::false_type{}, 0);
    for (int qt = 1; qt < nq; ++qt) tile(std::true_type{}, qt);
#pragma unroll
    for (int kt = 0; kt < W1_KT; ++kt) {
      const int key = kc * 256 + wave * 64 + kt * 32 + li;
      float* dkrow = dK + gbase + (size_t)key * ldg;
      float* dvrow = dV + gbase + (size_t)key * ldg;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(dkrow + 8 * g + 4 * lh) = make_float4(
            dkacc[kt][4 * g] * LN2, dkacc[kt][4 * g + 1] * LN2, dkacc[kt][4 * g + 2] * LN2, dkacc[kt][4 * g + 3] * LN2);
        *reinterpret_cast<float4*>(dvrow + 8 * g + 4 * lh) =
            make_float4(dvacc[kt][4 * g] * inv_keep, dvacc[kt][4 * g + 1] * inv_keep, dvacc[kt][4 * g + 2] * inv_keep,
                        dvacc[kt][4 * g + 3] * inv_keep);
      }
    }
  }
  {   // the last tile's dQ product
    float4 prev[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (nkc > 1) dq_prev((nq - 1) * 64, prev);
    dq_phase(nit - 1, prev, nkc > 1);
  }
#undef W1_LOAD_TILE
#undef W1_STORE
#undef W1_MFMA_PAIR
#undef W1_INTERLEAVE_12
}

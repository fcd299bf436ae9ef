// Bidirectional LSTM recurrence for the CRNN recognizer (model/crnn/crnn.py:6-22,66-68):
// nn.LSTM(nIn, 256, bidirectional), sequence-first, zero initial state, gate order (i,f,g,o)
// (SURVEY.md Appendix C).
//
// Split of the work:
//   * input projection  gx = x W_ih^T + b_ih  for all T at once: the implicit-GEMM kernel
//     (conv_igemm.hip) with a [rows, 2*4H] output (both directions side by side);
//   * the sequential part below: one launch per time step, both directions in the same grid
//     (blockIdx.z).  Block = 32 batch rows x 32 hidden units; wave g owns gate g and computes
//     h_{t-1}[32,H] . W_hh[g*H+unit, :]^T on fp32 MFMA straight from L2 (W_hh is 1 MiB per
//     direction and stays cache-resident across the T steps); gates meet in LDS.
//   * backward: per step  dh = dh_out + dgates_{t+1} . W_hh  (MFMA, K = 4H split over the 4
//     waves), then the cell derivative; d gx is stored for every t and the caller turns it
//     into dX with one GEMM (weights of a frozen recognizer need no dW).
//
// Layouts: gx, dgx [rows][2][4H] with row(t,b) = t*st_t + b*st_b (so the CNN's [B,T,C] output is
// consumed without a transpose); hseq, dhseq [T][B][2H]; gates [T][B][2][4H]; cseq [T][B][2][H].
#include "focr_common.h"
#include <type_traits>

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
// GRU scan: 48 transcendental gate evaluations per lane and time step sit on the critical path of a wave that owns
// its SIMD alone; the IEEE division and libm tanhf (~80 instructions per hidden unit) were most of the step time.
// v_exp + v_rcp forms: |error| < 2e-7 absolute (the tanh form cancels near 0, absolute error stays at rounding level).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(
    const float* __restrict__ gx, const float* __restrict__ whh, const float* __restrict__ bhh,
    float* __restrict__ hseq, float* __restrict__ gates, float* __restrict__ cseq, int step, int T,
    int B, int H, int st_t, int st_b) {
  __shared__ float pre[4][32][33];
  const int tid = threadIdx.x, g = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int dir = blockIdx.z, j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int t = dir == 0 ? step : T - 1 - step;
  const int tp = dir == 0 ? t - 1 : t + 1;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (step > 0) {
    int br = min(b0 + li, B - 1);
    const float* arow = hseq + ((size_t)tp * B + br) * 2 * H + dir * H + 4 * lh;
    const float* brow = whh + ((size_t)dir * 4 * H + g * H + j0 + li) * H + 4 * lh;
#pragma unroll 4
    for (int k = 0; k < H; k += 8) {
      float4 a = *reinterpret_cast<const float4*>(arow + k);
      float4 w = *reinterpret_cast<const float4*>(brow + k);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
  }
  {
    const int unit = j0 + li;
    const float bb = bhh[(size_t)dir * 4 * H + g * H + unit];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int bl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      int b = b0 + bl;
      float v = 0.f;
      if (b < B)
        v = acc[r] + bb + gx[((size_t)t * st_t + (size_t)b * st_b) * 8 * H + dir * 4 * H + g * H + unit];
      pre[g][bl][li] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int idx = tid + 256 * e;
    int bl = idx >> 5, u = idx & 31;
    int b = b0 + bl, unit = j0 + u;
    if (b >= B) continue;
    float ig = sigmoidf_(pre[0][bl][u]);
    float fg = sigmoidf_(pre[1][bl][u]);
    float gg = tanhf(pre[2][bl][u]);
    float og = sigmoidf_(pre[3][bl][u]);
    float cp = step > 0 ? cseq[(((size_t)tp * B + b) * 2 + dir) * H + unit] : 0.f;
    float c = fg * cp + ig * gg;
    float h = og * tanhf(c);
    size_t gb = (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
    gates[gb] = ig;
    gates[gb + H] = fg;
    gates[gb + 2 * H] = gg;
    gates[gb + 3 * H] = og;
    cseq[(((size_t)t * B + b) * 2 + dir) * H + unit] = c;
    hseq[((size_t)t * B + b) * 2 * H + dir * H + unit] = h;
  }
}

// backward step.  dgx rows use the same (st_t, st_b) mapping as gx.  dc_carry [2][B][H] holds
// dc_{t+1} * f_{t+1} between launches (written here for the next step).
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(
    const float* __restrict__ dhseq, const float* __restrict__ whh, const float* __restrict__ gates,
    const float* __restrict__ cseq, float* __restrict__ dgx, float* __restrict__ dc_carry, int step, int T,
    int B, int H, int st_t, int st_b) {
  __shared__ float part[4][32][33];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int dir = blockIdx.z, j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  // the backward walks each direction in the opposite order of its forward
  const int t = dir == 0 ? T - 1 - step : step;
  const int tn = dir == 0 ? t + 1 : t - 1;     // the step processed just before (later in fwd order)
  const int tp = dir == 0 ? t - 1 : t + 1;     // previous step of the forward recurrence
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (step > 0) {
    // dh_rec[b][unit] = sum_n dgate_{tn}[b][n] * Whh[n][unit], n in this wave's quarter of 4H
    int br = min(b0 + li, B - 1);
    const float* arow = dgx + ((size_t)tn * st_t + (size_t)br * st_b) * 8 * H + dir * 4 * H + wv * H + 4 * lh;
    const float* wbase = whh + ((size_t)dir * 4 * H + wv * H + 4 * lh) * H + j0 + li;
#pragma unroll 2
    for (int k = 0; k < H; k += 8) {
      float4 a = *reinterpret_cast<const float4*>(arow + k);
      const float* w = wbase + (size_t)k * H;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w[H], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w[2 * H], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w[3 * H], acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) part[wv][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int idx = tid + 256 * e;
    int bl = idx >> 5, u = idx & 31;
    int b = b0 + bl, unit = j0 + u;
    if (b >= B) continue;
    float dh = dhseq[((size_t)t * B + b) * 2 * H + dir * H + unit] + part[0][bl][u] + part[1][bl][u] +
               part[2][bl][u] + part[3][bl][u];
    size_t gb = (((size_t)t * B + b) * 2 + dir) * 4 * H + unit;
    float ig = gates[gb], fg = gates[gb + H], gg = gates[gb + 2 * H], og = gates[gb + 3 * H];
    float c = cseq[(((size_t)t * B + b) * 2 + dir) * H + unit];
    bool first = dir == 0 ? t == 0 : t == T - 1;
    float cp = first ? 0.f : cseq[(((size_t)tp * B + b) * 2 + dir) * H + unit];
    float tc = tanhf(c);
    size_t ci = ((size_t)dir * B + b) * H + unit;
    float dc = dh * og * (1.f - tc * tc) + (step > 0 ? dc_carry[ci] : 0.f);
    dc_carry[ci] = dc * fg;
    size_t ob = ((size_t)t * st_t + (size_t)b * st_b) * 8 * H + dir * 4 * H + unit;
    dgx[ob] = dc * gg * ig * (1.f - ig);
    dgx[ob + H] = dc * cp * fg * (1.f - fg);
    dgx[ob + 2 * H] = dc * ig * (1.f - gg * gg);
    dgx[ob + 3 * H] = dh * tc * og * (1.f - og);
  }
}

// whh: [2][4H][H] (forward direction then reverse), bhh: [2][4H]
int focr_lstm_fwd_bx3(const float* gx, const float* whh, const float* bhh, float* hseq, float* gates, float* cseq,
                      void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H, int st_t, int st_b,
                      hipStream_t stream);
int focr_lstm_bwd_bx3(const float* dhseq, const float* whh, const float* gates, const float* cseq, float* dgx,
                      float* dc_carry, void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H,
                      int st_t, int st_b, hipStream_t stream);
int focr_lstm_split_weights(const float* whh, void* out, int H, int backward, hipStream_t stream);

// ws: focr_lstm_ws_bytes(T,B,H,0) bytes (bf16 operand copies; only used in bf16x3 mode, may be null in fp32 mode)
// wsplit (the _pw entries; may be null): focr_lstm_split_bytes(H) bytes filled by focr_lstm_prepare_weights(whh, ...,
// backward) -- a caller whose recurrent weights do not change between calls (the frozen recognizer of the training step)
// prepares them once; the call then skips its weight-split launch.
extern "C" int focr_lstm_bidir_fwd_pw(const float* gx, const float* whh, const float* bhh, float* hseq,
                                      float* gates, float* cseq, void* ws, const void* wsplit, void* pflags,
                                      unsigned base, int T, int B, int H, int st_t, int st_b, hipStream_t stream) {
  FOCR_CHECK_ARG(gx && whh && bhh && hseq && gates && cseq, "null pointer");
  FOCR_CHECK_ARG(T > 0 && B > 0 && H % 32 == 0, "need H % 32 == 0");
  if (focr_get_precision() != 0 && ws && H == 256) {
    focr_lstm_fwd_bx3(gx, whh, bhh, hseq, gates, cseq, ws, wsplit, pflags, base, T, B, H, st_t, st_b, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  dim3 grid(H / 32, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, 256, 0, stream, gx, whh, bhh, hseq, gates, cseq, s, T, B, H,
                       st_t, st_b);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_lstm_bidir_fwd(const float* gx, const float* whh, const float* bhh, float* hseq,
                                   float* gates, float* cseq, void* ws, int T, int B, int H, int st_t,
                                   int st_b, hipStream_t stream) {
  return focr_lstm_bidir_fwd_pw(gx, whh, bhh, hseq, gates, cseq, ws, nullptr, nullptr, 0u, T, B, H, st_t, st_b, stream);
}

// dc_carry: 2*B*H floats of workspace.  dgx is fully overwritten.
extern "C" int focr_lstm_bidir_bwd_pw(const float* dhseq, const float* whh, const float* gates,
                                      const float* cseq, float* dgx, float* dc_carry, void* ws, const void* wsplit,
                                      void* pflags, unsigned base, int T, int B, int H, int st_t, int st_b,
                                      hipStream_t stream) {
  FOCR_CHECK_ARG(dhseq && whh && gates && cseq && dgx && dc_carry, "null pointer");
  FOCR_CHECK_ARG(T > 0 && B > 0 && H % 32 == 0, "need H % 32 == 0");
  if (focr_get_precision() != 0 && ws && H == 256) {
    focr_lstm_bwd_bx3(dhseq, whh, gates, cseq, dgx, dc_carry, ws, wsplit, pflags, base, T, B, H, st_t, st_b, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  dim3 grid(H / 32, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, 256, 0, stream, dhseq, whh, gates, cseq, dgx, dc_carry, s, T,
                       B, H, st_t, st_b);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_lstm_bidir_bwd(const float* dhseq, const float* whh, const float* gates,
                                   const float* cseq, float* dgx, float* dc_carry, void* ws, int T, int B,
                                   int H, int st_t, int st_b, hipStream_t stream) {
  return focr_lstm_bidir_bwd_pw(dhseq, whh, gates, cseq, dgx, dc_carry, ws, nullptr, nullptr, 0u, T, B, H, st_t, st_b, stream);
}
// 1 when focr_lstm_bidir_* would run this shape as ONE persistent launch (tuning key 2, residency, H): only then are the
// step counters of a caller-owned flag block touched (a caller that keeps a running `base` advances it only then).
static bool lp_usable(int B, int H);
extern "C" int focr_lstm_persistent_usable(int B, int H) {
  return (focr_get_precision() != 0 && H == 256 && lp_usable(B, H)) ? 1 : 0;
}
// hi / lo split of W_hh [2][4H][H] for the bf16x3 scans: backward = 0 -> [2][4H][H] (forward scan), 1 -> the transposed
// [2][H][4H] (backward scan); hi plane then lo plane, focr_lstm_split_bytes(H) bytes.
extern "C" long focr_lstm_split_bytes(int H) { return (long)2 * 2 * (2 * 4 * H * H); }
extern "C" int focr_lstm_prepare_weights(const float* whh, void* out, int H, int backward, hipStream_t stream) {
  FOCR_CHECK_ARG(whh && out && H > 0 && H % 32 == 0, "bad argument");
  const int rc = focr_lstm_split_weights(whh, out, H, backward, stream);
  FOCR_LAUNCH_CHECK();
  return rc;
}

// =======================================================================================
// Bidirectional GRU of the TSRN SRB (model/tsrn.py:128-145: nn.GRU(64, 32, bidirectional,
// batch_first), gate order (r,z,n), zero initial state).
//
// Wavefront scan: one wave owns 32 sequences of one direction for the WHOLE time loop -- one
// launch per GruBlock, no per-step launches, no LDS, no barriers.  The trick is the operand
// orientation of the f32 32x32x2 MFMA:  G^T[gate unit][seq] = sum_k W_hh[gate unit][k] h[seq][k]
//   A = W_hh rows (constant, 48 VGPRs for the 3 gates), B = h.
// The accumulator layout (lane = seq, register s <-> hidden unit u(s,half)) is exactly the B
// operand layout of the next step when the K index is enumerated as k = u(s,half), so the hidden
// state never leaves the accumulator registers between time steps.  The backward uses the same
// identity with A = W_hh^T.
// Map addressing: row(n,t) = (n/IC)*OS + (n%IC)*IS + t*TS  (see include/focr.h).
// =======================================================================================
#define GH 32
__device__ __forceinline__ int unit_of(int s, int lh) { return (s & 3) + 8 * (s >> 2) + 4 * lh; }

__global__ __launch_bounds__(256) void gru_fwd_kernel(const float* __restrict__ gx,
                                                      const float* __restrict__ whh,
                                                      const float* __restrict__ bhh,
                                                      float* __restrict__ hseq, float* __restrict__ gates,
                                                      int nseq, int T, int IC, int OS, int IS, int TS) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  float4 wa[3][4], bh[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      wa[g][q] = *reinterpret_cast<const float4*>(whh + ((size_t)dir * 96 + g * 32 + li) * GH + 8 * q + 4 * lh);
      bh[g][q] = *reinterpret_cast<const float4*>(bhh + (size_t)dir * 96 + g * 32 + 8 * q + 4 * lh);
    }
  f32x16 h;
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = 0.f;

  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    const long row = base_row + (long)t * TS;
    const float* gxr = gx + (size_t)row * 192 + dir * 96;
    float4 xg[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) xg[g][q] = *reinterpret_cast<const float4*>(gxr + g * 32 + 8 * q + 4 * lh);
    f32x16 ar, az, an;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ar[r] = 0.f; az[r] = 0.f; an[r] = 0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float wr[4] = {wa[0][q].x, wa[0][q].y, wa[0][q].z, wa[0][q].w};
      const float wz[4] = {wa[1][q].x, wa[1][q].y, wa[1][q].z, wa[1][q].w};
      const float wn[4] = {wa[2][q].x, wa[2][q].y, wa[2][q].z, wa[2][q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ar = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[e], h[4 * q + e], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_32x32x2f32(wz[e], h[4 * q + e], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_32x32x2f32(wn[e], h[4 * q + e], an, 0, 0, 0);
      }
    }
    float* grow = gates + ((size_t)row * 2 + dir) * 128;
    float* hrow = hseq + (size_t)row * 64 + dir * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xr[4] = {xg[0][q].x, xg[0][q].y, xg[0][q].z, xg[0][q].w};
      const float xz[4] = {xg[1][q].x, xg[1][q].y, xg[1][q].z, xg[1][q].w};
      const float xn[4] = {xg[2][q].x, xg[2][q].y, xg[2][q].z, xg[2][q].w};
      const float br[4] = {bh[0][q].x, bh[0][q].y, bh[0][q].z, bh[0][q].w};
      const float bz[4] = {bh[1][q].x, bh[1][q].y, bh[1][q].z, bh[1][q].w};
      const float bn[4] = {bh[2][q].x, bh[2][q].y, bh[2][q].z, bh[2][q].w};
      float rr[4], zz[4], nn[4], hn[4], hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int s = 4 * q + e;
        rr[e] = fast_sigmoid(xr[e] + ar[s] + br[e]);
        zz[e] = fast_sigmoid(xz[e] + az[s] + bz[e]);
        hn[e] = an[s] + bn[e];
        nn[e] = fast_tanh(xn[e] + rr[e] * hn[e]);
        hv[e] = (1.f - zz[e]) * nn[e] + zz[e] * h[s];
        h[s] = hv[e];
      }
      if (valid) {
        const int u = 8 * q + 4 * lh;
        *reinterpret_cast<float4*>(grow + u) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(grow + 32 + u) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        *reinterpret_cast<float4*>(grow + 64 + u) = make_float4(nn[0], nn[1], nn[2], nn[3]);
        *reinterpret_cast<float4*>(grow + 96 + u) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        *reinterpret_cast<float4*>(hrow + u) = make_float4(hv[0], hv[1], hv[2], hv[3]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ dhseq,
                                                      const float* __restrict__ whh,
                                                      const float* __restrict__ gates,
                                                      const float* __restrict__ hseq, float* __restrict__ dgx,
                                                      float* __restrict__ dgh, float* __restrict__ hprev,
                                                      int nseq, int T, int IC, int OS, int IS, int TS) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  // A operand of dh_prev^T[k][seq] = sum_j W_hh[j][k] dg[seq][j]:  lane (k = li), step s -> j = u(s,lh)
  float wt[3][16];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int s = 0; s < 16; ++s) wt[g][s] = whh[((size_t)dir * 96 + g * 32 + unit_of(s, lh)) * GH + li];
  f32x16 dh;
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] = 0.f;

  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;               // reverse of the forward order
    const bool has_prev = dir ? (t < T - 1) : (t > 0);
    const long row = base_row + (long)t * TS;
    const long prow = base_row + (long)(dir ? t + 1 : t - 1) * TS;
    const float* grow = gates + ((size_t)row * 2 + dir) * 128;
    float dar[16], daz[16], dhn[16], dhp[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = 8 * q + 4 * lh;
      float4 g4 = *reinterpret_cast<const float4*>(dhseq + (size_t)row * 64 + dir * 32 + u);
      float4 r4 = *reinterpret_cast<const float4*>(grow + u);
      float4 z4 = *reinterpret_cast<const float4*>(grow + 32 + u);
      float4 n4 = *reinterpret_cast<const float4*>(grow + 64 + u);
      float4 h4 = *reinterpret_cast<const float4*>(grow + 96 + u);
      float4 p4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_prev) p4 = *reinterpret_cast<const float4*>(hseq + (size_t)prow * 64 + dir * 32 + u);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
      const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
      const float hn[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {p4.x, p4.y, p4.z, p4.w};
      float o_r[4], o_z[4], o_n[4], o_h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int s = 4 * q + e;
        float dht = gg[e] + dh[s];
        float dn = dht * (1.f - zz[e]);
        float dz = dht * (hp[e] - nn[e]);
        dhp[s] = dht * zz[e];
        float dan = dn * (1.f - nn[e] * nn[e]);
        o_n[e] = dan;
        o_r[e] = dan * hn[e] * rr[e] * (1.f - rr[e]);
        o_h[e] = dan * rr[e];
        o_z[e] = dz * zz[e] * (1.f - zz[e]);
        dar[s] = o_r[e]; daz[s] = o_z[e]; dhn[s] = o_h[e];
      }
#if defined(GRU_ABL) && (GRU_ABL & 1)
      if (valid && step == T + 5) {
#else
      if (valid) {
#endif
        float* xo = dgx + (size_t)row * 192 + dir * 96;
        float* ho = dgh + (size_t)row * 192 + dir * 96;
        *reinterpret_cast<float4*>(xo + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(xo + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(xo + 64 + u) = make_float4(o_n[0], o_n[1], o_n[2], o_n[3]);
        *reinterpret_cast<float4*>(ho + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(ho + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(ho + 64 + u) = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
        *reinterpret_cast<float4*>(hprev + ((size_t)row * 2 + dir) * 32 + u) = p4;
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = dhp[r];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[0][s], dar[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[1][s], daz[s], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wt[2][s], dhn[s], acc, 0, 0, 0);
    }
    dh = acc;
  }
}

// ---------------------------------------------------------------------------------------
// bf16x3 variants (precision modes >= 1): the same wavefront scan with the recurrent product on the bf16 pipe.
// The f32 32x32x2 MFMA costs 16 passes per 2 k: 48 of them per step (3072 cycles) dominate the scan.  With split
// operands the step needs 3 gates x 2 k-steps x 3 products = 18 v_mfma_f32_32x32x16_bf16 (576 cycles).  The
// "state never leaves the registers" identity survives because the ORDER of the contraction index is free: k-slot
// (m, half, e) of the 16-wide fragment is defined as hidden unit u(8 m + e, half), which is exactly the unit held by
// accumulator register 8 m + e of the lane half -- a lane's B fragment is its own eight accumulator values (split to
// bf16 hi / lo in registers), and W_hh is laid out once per wave in the matching order.
// One wave per block (64 threads): the 2 * nseq / 32 waves of a launch spread over as many CUs as possible.
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8;
__device__ __forceinline__ void g_split8(const float* v, gbf16x8& hi, gbf16x8& lo) {
  const float x[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  focr_split8(x, hi, lo);
}
#define G_MFMA3(acc, ah, al, bh, bl)                                        \
  do {                                                                      \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);    \
  } while (0)

__global__ __launch_bounds__(64) void gru_fwd_bx3_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                         const float* __restrict__ bhh, float* __restrict__ hseq,
                                                         float* __restrict__ gates, int nseq, int T, int IC, int OS,
                                                         int IS, int TS) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  gbf16x8 wah[3][2], wal[3][2];            // A[gate unit li][k-slot (m, lh, e)] = W_hh[g*32 + li][u(8m + e, lh)]
  float4 bh[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + li) * GH + unit_of(8 * m + e, lh)];
      g_split8(v, wah[g][m], wal[g][m]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bh[g][q] = *reinterpret_cast<const float4*>(bhh + (size_t)dir * 96 + g * 32 + 8 * q + 4 * lh);
  }
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = 0.f;

  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    const long row = base_row + (long)t * TS;
    const float* gxr = gx + (size_t)row * 192 + dir * 96;
    float4 xg[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) xg[g][q] = *reinterpret_cast<const float4*>(gxr + g * 32 + 8 * q + 4 * lh);
    gbf16x8 hh[2], hl[2];
    g_split8(h, hh[0], hl[0]);
    g_split8(h + 8, hh[1], hl[1]);
    f32x16 ar, az, an;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ar[r] = 0.f; az[r] = 0.f; an[r] = 0.f; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      G_MFMA3(ar, wah[0][m], wal[0][m], hh[m], hl[m]);
      G_MFMA3(az, wah[1][m], wal[1][m], hh[m], hl[m]);
      G_MFMA3(an, wah[2][m], wal[2][m], hh[m], hl[m]);
    }
    float* grow = gates + ((size_t)row * 2 + dir) * 128;
    float* hrow = hseq + (size_t)row * 64 + dir * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xr[4] = {xg[0][q].x, xg[0][q].y, xg[0][q].z, xg[0][q].w};
      const float xz[4] = {xg[1][q].x, xg[1][q].y, xg[1][q].z, xg[1][q].w};
      const float xn[4] = {xg[2][q].x, xg[2][q].y, xg[2][q].z, xg[2][q].w};
      const float br[4] = {bh[0][q].x, bh[0][q].y, bh[0][q].z, bh[0][q].w};
      const float bz[4] = {bh[1][q].x, bh[1][q].y, bh[1][q].z, bh[1][q].w};
      const float bn[4] = {bh[2][q].x, bh[2][q].y, bh[2][q].z, bh[2][q].w};
      float rr[4], zz[4], nn[4], hn[4], hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * q + e;
        rr[e] = fast_sigmoid(xr[e] + ar[s] + br[e]);
        zz[e] = fast_sigmoid(xz[e] + az[s] + bz[e]);
        hn[e] = an[s] + bn[e];
        nn[e] = fast_tanh(xn[e] + rr[e] * hn[e]);
        hv[e] = (1.f - zz[e]) * nn[e] + zz[e] * h[s];
        h[s] = hv[e];
      }
      if (valid) {
        const int u = 8 * q + 4 * lh;
        *reinterpret_cast<float4*>(grow + u) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(grow + 32 + u) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        *reinterpret_cast<float4*>(grow + 64 + u) = make_float4(nn[0], nn[1], nn[2], nn[3]);
        *reinterpret_cast<float4*>(grow + 96 + u) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        *reinterpret_cast<float4*>(hrow + u) = make_float4(hv[0], hv[1], hv[2], hv[3]);
      }
    }
  }
}

__global__ __launch_bounds__(64) void gru_bwd_bx3_kernel(const float* __restrict__ dhseq, const float* __restrict__ whh,
                                                         const float* __restrict__ gates,
                                                         const float* __restrict__ hseq, float* __restrict__ dgx,
                                                         float* __restrict__ dgh, float* __restrict__ hprev, int nseq,
                                                         int T, int IC, int OS, int IS, int TS) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  // dh_prev^T[k][seq] = sum_j W_hh[j][k] dg[seq][j], j = (gate g, unit): A[k = li][k-slot (g, m, lh, e)] =
  // W_hh[g*32 + u(8m + e, lh)][li]; the B fragment of (g, m) = this lane's gate-gradient registers 8m .. 8m + 7
  gbf16x8 wth[3][2], wtl[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + unit_of(8 * m + e, lh)) * GH + li];
      g_split8(v, wth[g][m], wtl[g][m]);
    }
  float dh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] = 0.f;

  // operands of one step (24 float4): requested one step ahead, right before the MFMA chain of the current step
  float4 ld[4][6];
  auto request = [&](int step) {
    const int t = dir ? step : T - 1 - step;
    const bool has_prev = dir ? (t < T - 1) : (t > 0);
    const long row = base_row + (long)t * TS;
    const long prow = base_row + (long)(dir ? t + 1 : t - 1) * TS;
    const float* grow = gates + ((size_t)row * 2 + dir) * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = 8 * q + 4 * lh;
      ld[q][0] = *reinterpret_cast<const float4*>(dhseq + (size_t)row * 64 + dir * 32 + u);
      ld[q][1] = *reinterpret_cast<const float4*>(grow + u);
      ld[q][2] = *reinterpret_cast<const float4*>(grow + 32 + u);
      ld[q][3] = *reinterpret_cast<const float4*>(grow + 64 + u);
      ld[q][4] = *reinterpret_cast<const float4*>(grow + 96 + u);
      ld[q][5] = has_prev ? *reinterpret_cast<const float4*>(hseq + (size_t)prow * 64 + dir * 32 + u)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  request(0);
  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;               // reverse of the forward order
    const long row = base_row + (long)t * TS;
    float dar[16], daz[16], dhn[16], dhp[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = 8 * q + 4 * lh;
      const float4 g4 = ld[q][0], r4 = ld[q][1], z4 = ld[q][2], n4 = ld[q][3], h4 = ld[q][4], p4 = ld[q][5];
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
      const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
      const float hn[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {p4.x, p4.y, p4.z, p4.w};
      float o_r[4], o_z[4], o_n[4], o_h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * q + e;
        const float dht = gg[e] + dh[s];
        const float dn = dht * (1.f - zz[e]);
        const float dz = dht * (hp[e] - nn[e]);
        dhp[s] = dht * zz[e];
        const float dan = dn * (1.f - nn[e] * nn[e]);
        o_n[e] = dan;
        o_r[e] = dan * hn[e] * rr[e] * (1.f - rr[e]);
        o_h[e] = dan * rr[e];
        o_z[e] = dz * zz[e] * (1.f - zz[e]);
        dar[s] = o_r[e]; daz[s] = o_z[e]; dhn[s] = o_h[e];
      }
      if (valid) {
        float* xo = dgx + (size_t)row * 192 + dir * 96;
        float* ho = dgh + (size_t)row * 192 + dir * 96;
        *reinterpret_cast<float4*>(xo + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(xo + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(xo + 64 + u) = make_float4(o_n[0], o_n[1], o_n[2], o_n[3]);
        *reinterpret_cast<float4*>(ho + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(ho + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(ho + 64 + u) = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
        *reinterpret_cast<float4*>(hprev + ((size_t)row * 2 + dir) * 32 + u) = p4;
      }
    }
#if defined(GRU_ABL) && (GRU_ABL & 2)
    if (step + 1 < T && step < 1) request(step + 1);
#else
    if (step + 1 < T) request(step + 1);
#endif
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = dhp[r];
#if defined(GRU_ABL) && (GRU_ABL & 4)
    if (step == T + 5)
#endif
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      gbf16x8 bh_, bl_;
      g_split8(dar + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[0][m], wtl[0][m], bh_, bl_);
      g_split8(daz + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[1][m], wtl[1][m], bh_, bl_);
      g_split8(dhn + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[2][m], wtl[2][m], bh_, bl_);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------
// Loader / compute wave pairs (round 6; tuning key 5, default on).  The scans above are LATENCY-bound, not matrix-bound:
// the backward's per-step operands (24 x 16 bytes per lane: dh, r, z, n, hn, h_prev) were requested one step ahead into
// registers and still cost 1.8 of the 3.2 us per step standalone (tools/dev/gru_bench.py, -DGRU_ABL: 202 -> 87 us without
// the loads, T = 64) -- and 2.6x that inside the training step, where the weight-gradient stream keeps the memory system
// busy (525 us in the c1 trace).  A wave cannot keep more than 63 vector-memory operations in flight (vmcnt), and the
// compute wave's own stores count against that.  So the block gets a SECOND wave that does nothing but DMA the operands of
// steps t + 1 .. t + D - 1 straight into LDS (global_load_lds, 16 bytes per lane, no registers): its vmcnt holds only
// those transfers, the compute wave's only its stores.  One raw s_barrier per step hands a landed stage over (and the
// stage read in the previous step back).  LDS stage = [operand j][lane] x 16 bytes = the order the DMA writes and the
// ds_read_b128 of the same lane reads: linear, conflict-free.  Arithmetic and its order are those of the kernels above:
// results are bit-identical (tests/test_gpu_kernels.py::test_gru_loader_waves_equal_single_wave).
// ---------------------------------------------------------------------------------------
#define GL_BWD_OPS 24
#define GL_FWD_OPS 12
#define GL_LOADERS 2            // loader waves per block: each issues every second transfer of a stage, so each wave's vmcnt
                                // (63 at most) covers D - 1 stages in flight: 5 x 12 (backward) / 5 x 6 (forward)
#define GL_DMAX 6
__device__ __forceinline__ void gl_dma16(const float* g, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// Stage layout.  An operand row (32 floats of one sequence, one direction) is one 128-byte line in HBM; a transfer
// instruction moves 8 such lines: lane l fetches 16-byte chunk (l & 7) ^ (l >> 3) of the line of sequence 8 i + (l >> 3)
// (i = 0..3: four instructions per operand) and the DMA drops it at LDS offset 16 l of the instruction's 1 KB slot -- every
// instruction reads 8 whole lines (the register version read 32-byte pieces of 32 lines).  The compute lane (li, lh) finds
// chunk c = 2 q + lh of its sequence li at slot (li >> 3), row li & 7, position c ^ (li & 7): the XOR spreads the 8 lanes
// of a row group over different banks (rows are 128 bytes apart: unswizzled, a ds_read_b128 would be 8-way conflicted).
__device__ __forceinline__ float4 gl_lds16(const unsigned char* base, int w, int c, int li) {
  return *reinterpret_cast<const float4*>(base + (w * 4 + (li >> 3)) * 1024 + (li & 7) * 128 + ((c ^ (li & 7)) * 16));
}
// wait until at most `younger` stages of PER transfers each are outstanding (vmcnt is in order; a loader wave issues
// nothing else)
template <int PER>
__device__ __forceinline__ void gl_wait_stage(int younger) {
  switch (younger) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER) : "memory"); break;
  }
}

__global__ __launch_bounds__(64 * (1 + GL_LOADERS)) void gru_fwd_ld_kernel(
    const float* __restrict__ gx, const float* __restrict__ whh, const float* __restrict__ bhh, float* __restrict__ hseq,
    float* __restrict__ gates, int nseq, int T, int IC, int OS, int IS, int TS, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gl_ring[];      // D * GL_FWD_OPS * 1024 bytes
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int role = threadIdx.x >> 6;                    // 0 = compute, 1 .. GL_LOADERS = loaders
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  if (role >= 1) {
    // this lane's four source sequences (one per transfer instruction of an operand) and its chunk of their lines
    const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
    long brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sq = grp * 32 + 8 * i + r8;
      if (sq >= nseq) sq = nseq - 1;
      brow[i] = (long)(sq / IC) * OS + (long)(sq % IC) * IS;
    }
    auto run = [&](auto mec) {
      constexpr int me = decltype(mec)::value;           // (compile-time: every transfer's operand index is a constant)
      auto issue = [&](int step) {
        const int t = dir ? T - 1 - step : step;
        unsigned char* st = gl_ring + (step % D) * (GL_FWD_OPS * 1024);
#pragma unroll
        for (int j = 0; j < GL_FWD_OPS / GL_LOADERS; ++j) {
          const int o = j * GL_LOADERS + me, g = o >> 2, i = o & 3;
          gl_dma16(gx + (size_t)(brow[i] + (long)t * TS) * 192 + dir * 96 + g * 32 + 4 * chunk, st + o * 1024);
        }
      };
      for (int k = 0; k < D - 1; ++k)
        if (k < T) issue(k);
      for (int k = 0; k < T; ++k) {
        gl_wait_stage<GL_FWD_OPS / GL_LOADERS>(min(D - 2, T - 1 - k));
        __builtin_amdgcn_s_barrier();                   // hand-over k: stage k is in LDS; the compute wave is done with k - 1
        if (k + D - 1 < T) issue(k + D - 1);
      }
    };
    if (role == 1) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    return;
  }

  gbf16x8 wah[3][2], wal[3][2];
  float4 bh[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + li) * GH + unit_of(8 * m + e, lh)];
      g_split8(v, wah[g][m], wal[g][m]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bh[g][q] = *reinterpret_cast<const float4*>(bhh + (size_t)dir * 96 + g * 32 + 8 * q + 4 * lh);
  }
  float h[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) h[r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight loads: from here on this wave's vmcnt holds stores only

  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    const long row = base_row + (long)t * TS;
    // the recurrent product needs no operand of this step: it runs in front of the hand-over
    gbf16x8 hh[2], hl[2];
    g_split8(h, hh[0], hl[0]);
    g_split8(h + 8, hh[1], hl[1]);
    f32x16 ar, az, an;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ar[r] = 0.f; az[r] = 0.f; an[r] = 0.f; }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      G_MFMA3(ar, wah[0][m], wal[0][m], hh[m], hl[m]);
      G_MFMA3(az, wah[1][m], wal[1][m], hh[m], hl[m]);
      G_MFMA3(an, wah[2][m], wal[2][m], hh[m], hl[m]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the previous step's LDS reads are long consumed)
    __builtin_amdgcn_s_barrier();                       // hand-over `step`
    asm volatile("" ::: "memory");
    const unsigned char* st = gl_ring + (step % D) * (GL_FWD_OPS * 1024);
    float4 xg[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) xg[g][q] = gl_lds16(st, g, 2 * q + lh, li);
    float* grow = gates + ((size_t)row * 2 + dir) * 128;
    float* hrow = hseq + (size_t)row * 64 + dir * 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float xr[4] = {xg[0][q].x, xg[0][q].y, xg[0][q].z, xg[0][q].w};
      const float xz[4] = {xg[1][q].x, xg[1][q].y, xg[1][q].z, xg[1][q].w};
      const float xn[4] = {xg[2][q].x, xg[2][q].y, xg[2][q].z, xg[2][q].w};
      const float br[4] = {bh[0][q].x, bh[0][q].y, bh[0][q].z, bh[0][q].w};
      const float bz[4] = {bh[1][q].x, bh[1][q].y, bh[1][q].z, bh[1][q].w};
      const float bn[4] = {bh[2][q].x, bh[2][q].y, bh[2][q].z, bh[2][q].w};
      float rr[4], zz[4], nn[4], hn[4], hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * q + e;
        rr[e] = fast_sigmoid(xr[e] + ar[s] + br[e]);
        zz[e] = fast_sigmoid(xz[e] + az[s] + bz[e]);
        hn[e] = an[s] + bn[e];
        nn[e] = fast_tanh(xn[e] + rr[e] * hn[e]);
        hv[e] = (1.f - zz[e]) * nn[e] + zz[e] * h[s];
        h[s] = hv[e];
      }
      if (valid) {
        const int u = 8 * q + 4 * lh;
        *reinterpret_cast<float4*>(grow + u) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(grow + 32 + u) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        *reinterpret_cast<float4*>(grow + 64 + u) = make_float4(nn[0], nn[1], nn[2], nn[3]);
        *reinterpret_cast<float4*>(grow + 96 + u) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        *reinterpret_cast<float4*>(hrow + u) = make_float4(hv[0], hv[1], hv[2], hv[3]);
      }
    }
  }
}

__global__ __launch_bounds__(64 * (1 + GL_LOADERS)) void gru_bwd_ld_kernel(
    const float* __restrict__ dhseq, const float* __restrict__ whh, const float* __restrict__ gates,
    const float* __restrict__ hseq, float* __restrict__ dgx, float* __restrict__ dgh, float* __restrict__ hprev, int nseq,
    int T, int IC, int OS, int IS, int TS, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gl_ring[];      // D * GL_BWD_OPS * 1024 bytes
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int role = threadIdx.x >> 6;
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 32 >= nseq) return;
  const int seq = grp * 32 + li;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;

  if (role >= 1) {
    const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
    long brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sq = grp * 32 + 8 * i + r8;
      if (sq >= nseq) sq = nseq - 1;
      brow[i] = (long)(sq / IC) * OS + (long)(sq % IC) * IS;
    }
    auto run = [&](auto mec) {
      constexpr int me = decltype(mec)::value;
      auto issue = [&](int step) {
        const int t = dir ? step : T - 1 - step;
        const bool has_prev = dir ? (t < T - 1) : (t > 0);
        // (no previous step: the transfer still happens, from the step's own row, so that every stage is 24 transfers; the
        // compute wave substitutes zeros)
        const long dprev = has_prev ? (long)(dir ? 1 : -1) * TS : 0;
        unsigned char* st = gl_ring + (step % D) * (GL_BWD_OPS * 1024);
#pragma unroll
        for (int j = 0; j < GL_BWD_OPS / GL_LOADERS; ++j) {
          const int o = j * GL_LOADERS + me, w = o >> 2, i = o & 3;      // operand w: dh, r, z, n, hn, h_prev
          const long row = brow[i] + (long)t * TS;
          const float* src = w == 0 ? dhseq + (size_t)row * 64 + dir * 32
                             : w == 5 ? hseq + (size_t)(row + dprev) * 64 + dir * 32
                                      : gates + ((size_t)row * 2 + dir) * 128 + (w - 1) * 32;
          gl_dma16(src + 4 * chunk, st + o * 1024);
        }
      };
      for (int k = 0; k < D - 1; ++k)
        if (k < T) issue(k);
      for (int k = 0; k < T; ++k) {
        gl_wait_stage<GL_BWD_OPS / GL_LOADERS>(min(D - 2, T - 1 - k));
        __builtin_amdgcn_s_barrier();
        if (k + D - 1 < T) issue(k + D - 1);
      }
    };
    if (role == 1) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    return;
  }

  gbf16x8 wth[3][2], wtl[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + unit_of(8 * m + e, lh)) * GH + li];
      g_split8(v, wth[g][m], wtl[g][m]);
    }
  float dh[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) dh[r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;
    const bool has_prev = dir ? (t < T - 1) : (t > 0);
    const long row = base_row + (long)t * TS;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // hand-over `step`
    asm volatile("" ::: "memory");
    const unsigned char* st = gl_ring + (step % D) * (GL_BWD_OPS * 1024);
    float dar[16], daz[16], dhn[16], dhp[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = 8 * q + 4 * lh;
      const int c = 2 * q + lh;
      const float4 g4 = gl_lds16(st, 0, c, li), r4 = gl_lds16(st, 1, c, li), z4 = gl_lds16(st, 2, c, li);
      const float4 n4 = gl_lds16(st, 3, c, li), h4 = gl_lds16(st, 4, c, li);
      float4 p4 = gl_lds16(st, 5, c, li);
      if (!has_prev) p4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
      const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
      const float hn[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {p4.x, p4.y, p4.z, p4.w};
      float o_r[4], o_z[4], o_n[4], o_h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int s = 4 * q + e;
        const float dht = gg[e] + dh[s];
        const float dn = dht * (1.f - zz[e]);
        const float dz = dht * (hp[e] - nn[e]);
        dhp[s] = dht * zz[e];
        const float dan = dn * (1.f - nn[e] * nn[e]);
        o_n[e] = dan;
        o_r[e] = dan * hn[e] * rr[e] * (1.f - rr[e]);
        o_h[e] = dan * rr[e];
        o_z[e] = dz * zz[e] * (1.f - zz[e]);
        dar[s] = o_r[e]; daz[s] = o_z[e]; dhn[s] = o_h[e];
      }
      if (valid) {
        float* xo = dgx + (size_t)row * 192 + dir * 96;
        float* ho = dgh + (size_t)row * 192 + dir * 96;
        *reinterpret_cast<float4*>(xo + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(xo + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(xo + 64 + u) = make_float4(o_n[0], o_n[1], o_n[2], o_n[3]);
        *reinterpret_cast<float4*>(ho + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(ho + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(ho + 64 + u) = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
        *reinterpret_cast<float4*>(hprev + ((size_t)row * 2 + dir) * 32 + u) = p4;
      }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = dhp[r];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      gbf16x8 bh_, bl_;
      g_split8(dar + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[0][m], wtl[0][m], bh_, bl_);
      g_split8(daz + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[1][m], wtl[1][m], bh_, bl_);
      g_split8(dhn + 8 * m, bh_, bl_);
      G_MFMA3(acc, wth[2][m], wtl[2][m], bh_, bl_);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------
// 16-sequence waves (round 6, tuning key 5 = 2, the default).  With the loads out of the way the 32-sequence compute wave is
// ISSUE-bound: ~600 instructions per step for its 16 hidden units per lane, and the T = 64 scans of the TSRN blocks are only
// 128 such waves on a 256-CU chip.  On v_mfma_f32_16x16x32_bf16 a wave owns 16 sequences and a lane 8 hidden units -- the
// same "the accumulators ARE the next step's B fragment" identity holds: C/D lane (col = seq = l & 15, rows 4 g4 + j,
// g4 = l >> 4) of M-tile m is unit 16 m + 4 g4 + j, so k-slot (k-group g4, element e) of the 32-deep contraction is DEFINED
// as unit(g4, e) = e < 4 ? 4 g4 + e : 16 + 4 g4 + (e - 4) and the lane's own eight values are its B fragment; W_hh is
// gathered once per wave in that k order.  Twice the blocks, half the per-wave work, 18 quarter-size MFMAs per step; one
// loader wave suffices (12 / 6 whole-line transfers per stage: 5 stages in flight within vmcnt).  Same memory layouts as
// every other GRU kernel here; the contraction order inside an MFMA differs, so results agree with them to rounding, not
// bit for bit.
// ---------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float gf32x4;
#define G16_MFMA3(acc, ah, al, bh, bl)                                      \
  do {                                                                      \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);    \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);    \
  } while (0)
__device__ __forceinline__ int unit16(int kg, int e) { return e < 4 ? 4 * kg + e : 16 + 4 * kg + (e - 4); }
// chunk c (16 bytes) of operand w of sequence s (0..15) inside a stage: two 1 KB transfer slots per operand (8 sequences
// each), rows at 128-byte pitch, chunk position XORed with the row (see gl_lds16)
__device__ __forceinline__ float4 g16_lds(const unsigned char* base, int w, int c, int s) {
  return *reinterpret_cast<const float4*>(base + (w * 2 + (s >> 3)) * 1024 + (s & 7) * 128 + ((c ^ (s & 7)) * 16));
}
#define G16_FWD_OPS 6
#define G16_BWD_OPS 12

__global__ __launch_bounds__(128) void gru_fwd_l16_kernel(const float* __restrict__ gx, const float* __restrict__ whh,
                                                          const float* __restrict__ bhh, float* __restrict__ hseq,
                                                          float* __restrict__ gates, int nseq, int T, int IC, int OS,
                                                          int IS, int TS, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gl_ring[];      // D * G16_FWD_OPS * 1024 bytes
  const int lane = threadIdx.x & 63, sq = lane & 15, g4 = lane >> 4;
  const int role = threadIdx.x >> 6;
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 16 >= nseq) return;

  if (role == 1) {
    const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
    long brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int q = grp * 16 + 8 * i + r8;
      if (q >= nseq) q = nseq - 1;
      brow[i] = (long)(q / IC) * OS + (long)(q % IC) * IS;
    }
    auto issue = [&](int step) {
      const int t = dir ? T - 1 - step : step;
      unsigned char* st = gl_ring + (step % D) * (G16_FWD_OPS * 1024);
#pragma unroll
      for (int o = 0; o < G16_FWD_OPS; ++o) {
        const int g = o >> 1, i = o & 1;
        gl_dma16(gx + (size_t)(brow[i] + (long)t * TS) * 192 + dir * 96 + g * 32 + 4 * chunk, st + o * 1024);
      }
    };
    for (int k = 0; k < D - 1; ++k)
      if (k < T) issue(k);
    for (int k = 0; k < T; ++k) {
      gl_wait_stage<G16_FWD_OPS>(min(D - 2, T - 1 - k));
      __builtin_amdgcn_s_barrier();
      if (k + D - 1 < T) issue(k + D - 1);
    }
    return;
  }

  const int seq = grp * 16 + sq;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;
  gbf16x8 wah[3][2], wal[3][2];            // A[gate unit 16 m + (l & 15)][k-slot (g4, e)] = W_hh[g*32 + 16 m + r][unit16(g4, e)]
  float4 bh[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + 16 * m + sq) * GH + unit16(g4, e)];
      g_split8(v, wah[g][m], wal[g][m]);
      bh[g][m] = *reinterpret_cast<const float4*>(bhh + (size_t)dir * 96 + g * 32 + 16 * m + 4 * g4);
    }
  float h[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) h[r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int step = 0; step < T; ++step) {
    const int t = dir ? T - 1 - step : step;
    const long row = base_row + (long)t * TS;
    gbf16x8 hh, hl;
    g_split8(h, hh, hl);
    gf32x4 acc[3][2];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        acc[g][m] = gf32x4{0.f, 0.f, 0.f, 0.f};
        G16_MFMA3(acc[g][m], wah[g][m], wal[g][m], hh, hl);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // hand-over `step`
    asm volatile("" ::: "memory");
    const unsigned char* st = gl_ring + (step % D) * (G16_FWD_OPS * 1024);
    float* grow = gates + ((size_t)row * 2 + dir) * 128;
    float* hrow = hseq + (size_t)row * 64 + dir * 32;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int c = 4 * m + g4, u = 16 * m + 4 * g4;
      const float4 x0 = g16_lds(st, 0, c, sq), x1 = g16_lds(st, 1, c, sq), x2 = g16_lds(st, 2, c, sq);
      const float xr[4] = {x0.x, x0.y, x0.z, x0.w}, xz[4] = {x1.x, x1.y, x1.z, x1.w}, xn[4] = {x2.x, x2.y, x2.z, x2.w};
      const float br[4] = {bh[0][m].x, bh[0][m].y, bh[0][m].z, bh[0][m].w};
      const float bz[4] = {bh[1][m].x, bh[1][m].y, bh[1][m].z, bh[1][m].w};
      const float bn[4] = {bh[2][m].x, bh[2][m].y, bh[2][m].z, bh[2][m].w};
      float rr[4], zz[4], nn[4], hn[4], hv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * m + e;
        rr[e] = fast_sigmoid(xr[e] + acc[0][m][e] + br[e]);
        zz[e] = fast_sigmoid(xz[e] + acc[1][m][e] + bz[e]);
        hn[e] = acc[2][m][e] + bn[e];
        nn[e] = fast_tanh(xn[e] + rr[e] * hn[e]);
        hv[e] = (1.f - zz[e]) * nn[e] + zz[e] * h[i];
        h[i] = hv[e];
      }
      if (valid) {
        *reinterpret_cast<float4*>(grow + u) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4*>(grow + 32 + u) = make_float4(zz[0], zz[1], zz[2], zz[3]);
        *reinterpret_cast<float4*>(grow + 64 + u) = make_float4(nn[0], nn[1], nn[2], nn[3]);
        *reinterpret_cast<float4*>(grow + 96 + u) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        *reinterpret_cast<float4*>(hrow + u) = make_float4(hv[0], hv[1], hv[2], hv[3]);
      }
    }
  }
}

__global__ __launch_bounds__(128) void gru_bwd_l16_kernel(const float* __restrict__ dhseq, const float* __restrict__ whh,
                                                          const float* __restrict__ gates, const float* __restrict__ hseq,
                                                          float* __restrict__ dgx, float* __restrict__ dgh,
                                                          float* __restrict__ hprev, int nseq, int T, int IC, int OS,
                                                          int IS, int TS, int D) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gl_ring[];      // D * G16_BWD_OPS * 1024 bytes
  const int lane = threadIdx.x & 63, sq = lane & 15, g4 = lane >> 4;
  const int role = threadIdx.x >> 6;
  const int wid = blockIdx.x;
  const int dir = wid & 1, grp = wid >> 1;
  if (grp * 16 >= nseq) return;

  if (role == 1) {
    const int r8 = lane >> 3, chunk = (lane & 7) ^ r8;
    long brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int q = grp * 16 + 8 * i + r8;
      if (q >= nseq) q = nseq - 1;
      brow[i] = (long)(q / IC) * OS + (long)(q % IC) * IS;
    }
    auto issue = [&](int step) {
      const int t = dir ? step : T - 1 - step;
      const bool has_prev = dir ? (t < T - 1) : (t > 0);
      const long dprev = has_prev ? (long)(dir ? 1 : -1) * TS : 0;      // (no previous step: own row, the compute wave zeroes it)
      unsigned char* st = gl_ring + (step % D) * (G16_BWD_OPS * 1024);
#pragma unroll
      for (int o = 0; o < G16_BWD_OPS; ++o) {
        const int w = o >> 1, i = o & 1;                                   // operand w: dh, r, z, n, hn, h_prev
        const long row = brow[i] + (long)t * TS;
        const float* src = w == 0 ? dhseq + (size_t)row * 64 + dir * 32
                           : w == 5 ? hseq + (size_t)(row + dprev) * 64 + dir * 32
                                    : gates + ((size_t)row * 2 + dir) * 128 + (w - 1) * 32;
        gl_dma16(src + 4 * chunk, st + o * 1024);
      }
    };
    for (int k = 0; k < D - 1; ++k)
      if (k < T) issue(k);
    for (int k = 0; k < T; ++k) {
      gl_wait_stage<G16_BWD_OPS>(min(D - 2, T - 1 - k));
      __builtin_amdgcn_s_barrier();
      if (k + D - 1 < T) issue(k + D - 1);
    }
    return;
  }

  const int seq = grp * 16 + sq;
  const bool valid = seq < nseq;
  const int sc = valid ? seq : nseq - 1;
  const long base_row = (long)(sc / IC) * OS + (long)(sc % IC) * IS;
  // dh_prev^T[k][seq] = sum_j W_hh[j][k] dg[seq][j]: A[k = 16 m + (l & 15)][k-slot (g4, e) of gate g] = W_hh[g*32 + unit16(g4, e)][k]
  gbf16x8 wth[3][2], wtl[3][2];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = whh[((size_t)dir * 96 + g * 32 + unit16(g4, e)) * GH + 16 * m + sq];
      g_split8(v, wth[g][m], wtl[g][m]);
    }
  float dh[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) dh[r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int step = 0; step < T; ++step) {
    const int t = dir ? step : T - 1 - step;
    const bool has_prev = dir ? (t < T - 1) : (t > 0);
    const long row = base_row + (long)t * TS;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // hand-over `step`
    asm volatile("" ::: "memory");
    const unsigned char* st = gl_ring + (step % D) * (G16_BWD_OPS * 1024);
    float dar[8], daz[8], dhn[8], dhp[8];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int c = 4 * m + g4, u = 16 * m + 4 * g4;
      const float4 g4v = g16_lds(st, 0, c, sq), r4 = g16_lds(st, 1, c, sq), z4 = g16_lds(st, 2, c, sq);
      const float4 n4 = g16_lds(st, 3, c, sq), h4 = g16_lds(st, 4, c, sq);
      float4 p4 = g16_lds(st, 5, c, sq);
      if (!has_prev) p4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float gg[4] = {g4v.x, g4v.y, g4v.z, g4v.w}, rr[4] = {r4.x, r4.y, r4.z, r4.w};
      const float zz[4] = {z4.x, z4.y, z4.z, z4.w}, nn[4] = {n4.x, n4.y, n4.z, n4.w};
      const float hn[4] = {h4.x, h4.y, h4.z, h4.w}, hp[4] = {p4.x, p4.y, p4.z, p4.w};
      float o_r[4], o_z[4], o_n[4], o_h[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * m + e;
        const float dht = gg[e] + dh[i];
        const float dn = dht * (1.f - zz[e]);
        const float dz = dht * (hp[e] - nn[e]);
        dhp[i] = dht * zz[e];
        const float dan = dn * (1.f - nn[e] * nn[e]);
        o_n[e] = dan;
        o_r[e] = dan * hn[e] * rr[e] * (1.f - rr[e]);
        o_h[e] = dan * rr[e];
        o_z[e] = dz * zz[e] * (1.f - zz[e]);
        dar[i] = o_r[e]; daz[i] = o_z[e]; dhn[i] = o_h[e];
      }
      if (valid) {
        float* xo = dgx + (size_t)row * 192 + dir * 96;
        float* ho = dgh + (size_t)row * 192 + dir * 96;
        *reinterpret_cast<float4*>(xo + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(xo + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(xo + 64 + u) = make_float4(o_n[0], o_n[1], o_n[2], o_n[3]);
        *reinterpret_cast<float4*>(ho + u) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
        *reinterpret_cast<float4*>(ho + 32 + u) = make_float4(o_z[0], o_z[1], o_z[2], o_z[3]);
        *reinterpret_cast<float4*>(ho + 64 + u) = make_float4(o_h[0], o_h[1], o_h[2], o_h[3]);
        *reinterpret_cast<float4*>(hprev + ((size_t)row * 2 + dir) * 32 + u) = p4;
      }
    }
    gf32x4 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[m] = gf32x4{dhp[4 * m], dhp[4 * m + 1], dhp[4 * m + 2], dhp[4 * m + 3]};
    gbf16x8 bh_, bl_;
    g_split8(dar, bh_, bl_);
    G16_MFMA3(acc[0], wth[0][0], wtl[0][0], bh_, bl_);
    G16_MFMA3(acc[1], wth[0][1], wtl[0][1], bh_, bl_);
    g_split8(daz, bh_, bl_);
    G16_MFMA3(acc[0], wth[1][0], wtl[1][0], bh_, bl_);
    G16_MFMA3(acc[1], wth[1][1], wtl[1][1], bh_, bl_);
    g_split8(dhn, bh_, bl_);
    G16_MFMA3(acc[0], wth[2][0], wtl[2][0], bh_, bl_);
    G16_MFMA3(acc[1], wth[2][1], wtl[2][1], bh_, bl_);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[4 * m + e] = acc[m][e];
  }
}

// ring depth for a scan of T steps over `blocks` blocks: deep rings (fewer resident blocks per CU) where the launch does
// not fill the chip anyway or the scan is long; the smallest ring that still keeps two stages in flight for short scans
static int gl_ring_depth(int T, int blocks, int ops) {
  int d = (T >= 32 || blocks <= 256) ? GL_DMAX : 3;
  if (d > T + 1) d = T + 1 < 2 ? 2 : T + 1;
  while (d > 2 && (size_t)d * ops * 1024 > 150 * 1024) --d;
  return d;
}
template <typename KernT>
static bool gl_set_lds(KernT kern, size_t lds, focr_dev_flags& attr) {
  if (focr_dev_first(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            GL_DMAX * GL_BWD_OPS * 1024) != hipSuccess)
      return false;
    focr_dev_mark(attr);
  }
  (void)lds;
  return true;
}

// dW_hh of both directions from the [192 x 64] cross product G = dgh^T [h_prev(dir 0) | h_prev(dir 1)] that ONE streaming
// weight-gradient launch produces (linear_wgrad.hip, QUART mode): dW_hh[d][j][k] = G[96 d + j][32 d + k]; the off-diagonal
// blocks (gate gradients of one direction against the other direction's state) are discarded.
// cross also carries, behind the matrix, the 192 column sums of dgh (the bias output of the same launch) = db_hh of both
// directions.
__global__ __launch_bounds__(256) void gru_whh_extract_kernel(const float* __restrict__ G, float* __restrict__ dwhh,
                                                              float* __restrict__ dbhh, int accumulate) {
  const int i = blockIdx.x * 256 + threadIdx.x;          // 2 * 96 * 32 = 6144 matrix elements + 192 bias elements
  if (i < 6144) {
    const int d = i / 3072, r = i - d * 3072, j = r >> 5, k = r & 31;
    const float v = G[(96 * d + j) * 64 + 32 * d + k];
    dwhh[i] = accumulate ? dwhh[i] + v : v;
  } else if (i < 6144 + 192 && dbhh) {
    const float v = G[192 * 64 + (i - 6144)];
    dbhh[i - 6144] = accumulate ? dbhh[i - 6144] + v : v;
  }
}
extern "C" int focr_gru_whh_extract(const float* cross, float* dwhh, float* dbhh, int accumulate, hipStream_t stream) {
  FOCR_CHECK_ARG(cross && dwhh, "null pointer");
  hipLaunchKernelGGL(gru_whh_extract_kernel, dim3(25), 256, 0, stream, cross, dwhh, dbhh, accumulate);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_gru_bidir_fwd(const float* gx, const float* whh, const float* bhh, float* hseq,
                                  float* gates, int nseq, int T, int IC, int OS, int IS, int TS,
                                  hipStream_t stream) {
  FOCR_CHECK_ARG(gx && whh && bhh && hseq && gates, "null pointer");
  FOCR_CHECK_ARG(nseq > 0 && T > 0 && IC > 0, "bad argument");
  int waves = cdiv(nseq, 32) * 2;
  if (focr_get_precision() != 0 && focr_get_tuning(FOCR_TUNE_GRU_LOADER) >= 2) {
    const int blocks = cdiv(nseq, 16) * 2;
    int D = blocks > 512 ? 3 : GL_DMAX;
    if (D > T + 1) D = T + 1;
    const size_t lds = (size_t)D * G16_FWD_OPS * 1024;
    static focr_dev_flags attr;
    if (!gl_set_lds(gru_fwd_l16_kernel, lds, attr)) {
      focr_set_error("focr_gru_bidir_fwd: cannot reserve %zu bytes of LDS", lds);
      return FOCR_EHIP;
    }
    hipLaunchKernelGGL(gru_fwd_l16_kernel, dim3(blocks), 128, lds, stream, gx, whh, bhh, hseq, gates, nseq, T, IC, OS, IS, TS, D);
  } else if (focr_get_precision() != 0 && focr_get_tuning(FOCR_TUNE_GRU_LOADER) != 0) {
    const int D = gl_ring_depth(T, waves, GL_FWD_OPS);
    const size_t lds = (size_t)D * GL_FWD_OPS * 1024;
    static focr_dev_flags attr;
    if (!gl_set_lds(gru_fwd_ld_kernel, lds, attr)) {
      focr_set_error("focr_gru_bidir_fwd: cannot reserve %zu bytes of LDS", lds);
      return FOCR_EHIP;
    }
    hipLaunchKernelGGL(gru_fwd_ld_kernel, dim3(waves), 64 * (1 + GL_LOADERS), lds, stream, gx, whh, bhh, hseq, gates, nseq, T,
                       IC, OS, IS, TS, D);
  }
  else if (focr_get_precision() != 0)
    hipLaunchKernelGGL(gru_fwd_bx3_kernel, dim3(waves), 64, 0, stream, gx, whh, bhh, hseq, gates, nseq, T, IC, OS, IS, TS);
  else
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(cdiv(waves, 4)), 256, 0, stream, gx, whh, bhh, hseq, gates, nseq, T, IC,
                       OS, IS, TS);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_gru_bidir_bwd(const float* dhseq, const float* whh, const float* gates,
                                  const float* hseq, float* dgx, float* dgh, float* hprev, int nseq, int T,
                                  int IC, int OS, int IS, int TS, hipStream_t stream) {
  FOCR_CHECK_ARG(dhseq && whh && gates && hseq && dgx && dgh && hprev, "null pointer");
  FOCR_CHECK_ARG(nseq > 0 && T > 0 && IC > 0, "bad argument");
  int waves = cdiv(nseq, 32) * 2;
  if (focr_get_precision() != 0 && focr_get_tuning(FOCR_TUNE_GRU_LOADER) >= 2) {
    const int blocks = cdiv(nseq, 16) * 2;
    int D = blocks > 512 ? 3 : GL_DMAX;
    if (D > T + 1) D = T + 1;
    const size_t lds = (size_t)D * G16_BWD_OPS * 1024;
    static focr_dev_flags attr;
    if (!gl_set_lds(gru_bwd_l16_kernel, lds, attr)) {
      focr_set_error("focr_gru_bidir_bwd: cannot reserve %zu bytes of LDS", lds);
      return FOCR_EHIP;
    }
    hipLaunchKernelGGL(gru_bwd_l16_kernel, dim3(blocks), 128, lds, stream, dhseq, whh, gates, hseq, dgx, dgh, hprev, nseq, T,
                       IC, OS, IS, TS, D);
  } else if (focr_get_precision() != 0 && focr_get_tuning(FOCR_TUNE_GRU_LOADER) != 0) {
    const int D = gl_ring_depth(T, waves, GL_BWD_OPS);
    const size_t lds = (size_t)D * GL_BWD_OPS * 1024;
    static focr_dev_flags attr;
    if (!gl_set_lds(gru_bwd_ld_kernel, lds, attr)) {
      focr_set_error("focr_gru_bidir_bwd: cannot reserve %zu bytes of LDS", lds);
      return FOCR_EHIP;
    }
    hipLaunchKernelGGL(gru_bwd_ld_kernel, dim3(waves), 64 * (1 + GL_LOADERS), lds, stream, dhseq, whh, gates, hseq, dgx, dgh,
                       hprev, nseq, T, IC, OS, IS, TS, D);
  } else if (focr_get_precision() != 0)
    hipLaunchKernelGGL(gru_bwd_bx3_kernel, dim3(waves), 64, 0, stream, dhseq, whh, gates, hseq, dgx, dgh, hprev, nseq, T,
                       IC, OS, IS, TS);
  else
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(cdiv(waves, 4)), 256, 0, stream, dhseq, whh, gates, hseq, dgx, dgh,
                       hprev, nseq, T, IC, OS, IS, TS);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// =======================================================================================
// bf16x3 variants of the LSTM step kernels (precision mode 1).
// The recurrent GEMM h_{t-1} W_hh^T (K = 256) runs on v_mfma_f32_32x32x16_bf16 with split operands:
//   * W_hh (frozen recogniser weights) is split ONCE into bf16 hi/lo: whh2[2(hi,lo)][2][4H][H], and for
//     the backward also transposed: whhT2[2][2][H][4H];
//   * every step writes its h (forward) / gate gradients (backward) also as bf16 hi/lo side buffers, so
//     the next step's A fragments are plain 16-byte loads (no per-step conversion of the operand).
// 48 MFMAs (1536 cycles) per wave and step instead of 128 f32 MFMAs (8192 cycles).
// =======================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 rbf16x8;

// dst2[0] = hi, dst2[1] = lo of src (n elements); optional transpose of [rows][cols] matrices (count of them = mats)
__global__ void split_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n, int rows, int cols,
                                  int transpose) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    long o = i;
    if (transpose) {
      long per = (long)rows * cols;
      long m = i / per, r = (i % per) / cols, c = i % cols;
      o = m * per + c * rows + r;
    }
    float v = src[i];
    __bf16 h = (__bf16)v;
    dst[o] = h;
    dst[n + o] = (__bf16)(v - (float)h);
  }
}

// Block = 16 waves: wave (g = wave & 3, kq = wave >> 2) computes gate g's partial product over the K quarter
// [64 kq, 64 kq + 64) with all 16 operand loads in flight at once (the step is bound by load latency: h was
// written by other XCDs in the previous launch and comes from MALL/HBM).  Partials are folded in a FIXED order
// through LDS (bit-deterministic forward); the cell inputs gx / c_{t-1} / bias are prefetched before the GEMM.
#define LSTM_MFMA3_QUARTER(arow, brow, aoff, boff)                                                      \
  {                                                                                                     \
    rbf16x8 ah[4], al[4], wh[4], wl[4];                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      ah[i] = *reinterpret_cast<const rbf16x8*>(arow + 16 * i);                                         \
      al[i] = *reinterpret_cast<const rbf16x8*>(arow + aoff + 16 * i);                                  \
      wh[i] = *reinterpret_cast<const rbf16x8*>(brow + 16 * i);                                         \
      wl[i] = *reinterpret_cast<const rbf16x8*>(brow + boff + 16 * i);                                  \
    }                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wh[i], acc, 0, 0, 0);                        \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], wl[i], acc, 0, 0, 0);                        \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], wh[i], acc, 0, 0, 0);                        \
    }                                                                                                   \
  }

__global__ __launch_bounds__(1024) void lstm_fwd_step_bx3_kernel(
    const float* __restrict__ gx, const __bf16* __restrict__ whh2, const float* __restrict__ bhh,
    float* __restrict__ hseq, __bf16* __restrict__ hseq2, float* __restrict__ gates, float* __restrict__ cseq,
    int step, int T, int B, int H, int st_t, int st_b) {
  __shared__ float part[3][4][32][33];
  const int tid = threadIdx.x, wave = tid >> 6, g = wave & 3, kq = wave >> 2;
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int dir = blockIdx.z, j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int t = dir == 0 ? step : T - 1 - step;
  const int tp = dir == 0 ? t - 1 : t + 1;
  const long nh = (long)T * B * 2 * H;        // elements of hseq (offset of the lo plane in hseq2)
  const long nw = (long)2 * 4 * H * H;        // elements of whh
  // cell element owned by this thread in the epilogue
  const int ebl = tid >> 5, eu = tid & 31, eb = b0 + ebl, eunit = j0 + eu;
  float gxv[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
  if (eb < B) {
    const float* gp = gx + ((size_t)t * st_t + (size_t)eb * st_b) * 8 * H + dir * 4 * H + eunit;
    const float* bp = bhh + (size_t)dir * 4 * H + eunit;
#pragma unroll
    for (int q = 0; q < 4; ++q) gxv[q] = gp[q * H] + bp[q * H];
    if (step > 0) cp = cseq[(((size_t)tp * B + eb) * 2 + dir) * H + eunit];
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (step > 0) {
    int br = min(b0 + li, B - 1);
    const __bf16* arow = hseq2 + ((size_t)tp * B + br) * 2 * H + dir * H + 64 * kq + 8 * lh;
    const __bf16* brow = whh2 + ((size_t)dir * 4 * H + g * H + j0 + li) * H + 64 * kq + 8 * lh;
    LSTM_MFMA3_QUARTER(arow, brow, nh, nw)
  }
  if (kq > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[kq - 1][g][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  }
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int bl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      part[0][g][bl][li] = ((acc[r] + part[0][g][bl][li]) + part[1][g][bl][li]) + part[2][g][bl][li];
    }
  }
  __syncthreads();
  if (eb < B) {
    float ig = sigmoidf_(part[0][0][ebl][eu] + gxv[0]);
    float fg = sigmoidf_(part[0][1][ebl][eu] + gxv[1]);
    float gg = tanhf(part[0][2][ebl][eu] + gxv[2]);
    float og = sigmoidf_(part[0][3][ebl][eu] + gxv[3]);
    float c = fg * cp + ig * gg;
    float h = og * tanhf(c);
    size_t gb = (((size_t)t * B + eb) * 2 + dir) * 4 * H + eunit;
    gates[gb] = ig;
    gates[gb + H] = fg;
    gates[gb + 2 * H] = gg;
    gates[gb + 3 * H] = og;
    cseq[(((size_t)t * B + eb) * 2 + dir) * H + eunit] = c;
    size_t ho = ((size_t)t * B + eb) * 2 * H + dir * H + eunit;
    hseq[ho] = h;
    __bf16 hh = (__bf16)h;
    hseq2[ho] = hh;
    hseq2[nh + ho] = (__bf16)(h - (float)hh);
  }
}

// dgx2: bf16 hi/lo copy of dgx ([rows][2][4H], same row mapping), written here for the next step.
// wave (nq = wave & 3, kq = wave >> 2) covers the reduction index n in [nq H + 64 kq, + 64) (H == 256).
__global__ __launch_bounds__(1024) void lstm_bwd_step_bx3_kernel(
    const float* __restrict__ dhseq, const __bf16* __restrict__ whhT2, const float* __restrict__ gates,
    const float* __restrict__ cseq, float* __restrict__ dgx, __bf16* __restrict__ dgx2, float* __restrict__ dc_carry,
    int step, int T, int B, int H, int st_t, int st_b, long ndg) {
  __shared__ float part[3][4][32][33];
  const int tid = threadIdx.x, wave = tid >> 6, nq = wave & 3, kq = wave >> 2;
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int dir = blockIdx.z, j0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int t = dir == 0 ? T - 1 - step : step;
  const int tn = dir == 0 ? t + 1 : t - 1;
  const int tp = dir == 0 ? t - 1 : t + 1;
  const long nwt = (long)2 * H * 4 * H;
  const int ebl = tid >> 5, eu = tid & 31, eb = b0 + ebl, eunit = j0 + eu;
  const bool first = dir == 0 ? t == 0 : t == T - 1;
  float dh0 = 0.f, ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, cp = 0.f, carry = 0.f;
  const size_t ci = ((size_t)dir * B + eb) * H + eunit;
  if (eb < B) {
    dh0 = dhseq[((size_t)t * B + eb) * 2 * H + dir * H + eunit];
    size_t gb = (((size_t)t * B + eb) * 2 + dir) * 4 * H + eunit;
    ig = gates[gb], fg = gates[gb + H], gg = gates[gb + 2 * H], og = gates[gb + 3 * H];
    c = cseq[(((size_t)t * B + eb) * 2 + dir) * H + eunit];
    cp = first ? 0.f : cseq[(((size_t)tp * B + eb) * 2 + dir) * H + eunit];
    carry = step > 0 ? dc_carry[ci] : 0.f;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (step > 0) {
    int br = min(b0 + li, B - 1);
    const __bf16* arow =
        dgx2 + ((size_t)tn * st_t + (size_t)br * st_b) * 8 * H + dir * 4 * H + nq * H + 64 * kq + 8 * lh;
    const __bf16* brow = whhT2 + ((size_t)dir * H + j0 + li) * 4 * H + nq * H + 64 * kq + 8 * lh;   // WhhT[unit][n]
    LSTM_MFMA3_QUARTER(arow, brow, ndg, nwt)
  }
  if (kq > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[kq - 1][nq][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
  }
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int bl = (r & 3) + 8 * (r >> 2) + 4 * lh;
      part[0][nq][bl][li] = ((acc[r] + part[0][nq][bl][li]) + part[1][nq][bl][li]) + part[2][nq][bl][li];
    }
  }
  __syncthreads();
  if (eb < B) {
    float dh = dh0 + (((part[0][0][ebl][eu] + part[0][1][ebl][eu]) + part[0][2][ebl][eu]) + part[0][3][ebl][eu]);
    float tc = tanhf(c);
    float dc = dh * og * (1.f - tc * tc) + carry;
    dc_carry[ci] = dc * fg;
    size_t ob = ((size_t)t * st_t + (size_t)eb * st_b) * 8 * H + dir * 4 * H + eunit;
    float d4[4] = {dc * gg * ig * (1.f - ig), dc * cp * fg * (1.f - fg), dc * ig * (1.f - gg * gg),
                   dh * tc * og * (1.f - og)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dgx[ob + q * H] = d4[q];
      __bf16 hh = (__bf16)d4[q];
      dgx2[ob + q * H] = hh;
      dgx2[ndg + ob + q * H] = (__bf16)(d4[q] - (float)hh);
    }
  }
}

// =======================================================================================
// Persistent variants: ONE launch walks all T steps (a BiLSTM layer = 1 launch instead of T).
//   * block (jx, by, dir) owns 32 hidden units x 32 sequences of one direction for the whole scan; its slice of
//     W_hh (4 gates x 32 units x 256, bf16 hi + lo = 128 KB) is read ONCE into LDS and stays there;
//   * c_{t-1} / the carried dc live in registers of the thread that owns the cell element (ownership is fixed);
//   * step t needs h_{t-1} of ALL 256 units of its 32 sequences = the output of the 8 blocks (jx = 0..7) that share
//     (by, dir): those 8 blocks meet at a counter in global memory once per step -- release increment after the block's
//     h / gate-gradient rows are written, acquire spin before they are read (agent scope: the 8 blocks may sit on
//     different XCDs, whose L2s are not coherent with each other).  Block ids are laid out so that the 8 partners
//     have the same id mod 8 (= the same XCD under round-robin dispatch), which keeps the exchange inside one L2;
//   * the inputs of the cell update that do not depend on the partners (gx, bias; gates, c, dh in the backward) are
//     requested BEFORE the wait;
//   * a bounded spin (LSTM_SPIN_LIMIT polls) turns a scheduling surprise into an error word instead of a hung GPU.
// All 8 x (B/32) x 2 blocks must be resident at once (64 blocks of 512 threads / 152 KB LDS at B = 128; the launcher
// falls back to the per-step kernels beyond 256 blocks).
// =======================================================================================
#define LP_WP 264                       // bf16 pitch of a 256-long W row (528 B: conflict-free ds_read_b128)
#define LP_WTP 1032                     // bf16 pitch of a 1024-long W^T row
#ifndef LSTM_SPIN_LIMIT
#define LSTM_SPIN_LIMIT (1 << 22)
#endif

// `bad` (LDS, zeroed by the kernel): set when the wait timed out, i.e. a partner block was never scheduled.  The kernels
// then write NaN into every later output of this block, so a scan that could not synchronise can never pass for a
// result (the error word `err` additionally tells the host why; kernels._lstm_check reads it under FOCR_LSTM_CHECK=1).
// Exchange protocol (round 4 re-measured both forms of cdna_hip_programming.md Guideline 16 on this scan):
//   LSTM_XCHG == 0 (default): payload = plain stores (they stay in the XCD's L2, where the 8 partners -- same id mod 8 =
//     same XCD under round-robin dispatch -- read them back), ONE lane bumps the group's step counter with RELEASE order
//     (agent scope: correct on any placement), the consumer polls the word RELAXED and issues ONE agent-scope acquire after
//     the match (round 3 polled with ACQUIRE loads: an L1 invalidate per poll), then plain 16-byte loads.
//   LSTM_XCHG == 1: the fence-free form -- 8-byte agent-scope atomic stores (write-through) and loads (past the L1) on
//     both sides, relaxed counter.  Correct (bit-identical scan), but SLOWER here: 250 / 265 us per layer against 207 / 213:
//     write-through stores drop the line from the L2 the partners would have hit, and 8-byte accesses run at 0.5-0.7x the
//     16-byte rate (MI355X_MICROARCH.md, visibility table).
//   Also tried: plain stores + drained RELAXED counter + L1-bypassing loads and no fence at all (sound only if the partners
//   share an L2): the scan goes STALE (test_lstm / decoded strings fail) and is slower still (280 / 316 us) -- dropped.
// In both forms the payload is staged in LDS and leaves as 8-byte stores (round 3: 2-byte stores, 16 per thread in the
// backward scan).  Net effect of the round-4 form: 207 / 213 -> 202 / 216 us per layer, i.e. none: the 8 us per step are not
// in the polls or the store width.
#ifndef LSTM_XCHG
#define LSTM_XCHG 0
#endif
__device__ __forceinline__ void lp_wait(unsigned* flag, unsigned target, unsigned* err, unsigned* bad) {
#ifdef LSTM_ABL_NOSYNC
  __syncthreads();
  return;
#endif
  if (threadIdx.x == 0) {
    int spins = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {     // (wrap-safe)
      if (++spins > LSTM_SPIN_LIMIT) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *bad = 1u;
        break;
      }
    }
#if LSTM_XCHG == 0
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // ONE invalidate of this CU's L1, after the match
#endif
  }
  __syncthreads();
}
// fast (round 5, wave-uniform per block): ALL 8 partners of this block's group run on ITS XCD (checked at run time, see
// lp_same_xcd below).  The payload then only has to be in the XCD's L2 before the counter moves: every storing wave waits
// for its stores' acknowledgement (vmcnt(0)), barrier, RELAXED increment -- the workgroup-scope release of a
// threadgroup-split workgroup (CUs behind one L2) -- instead of the agent-scope release, whose L2 write-back was 1.1-1.3 us
// of every step (profiles/r05_lstm_phases_*.txt).  The consumer side is unchanged (relaxed poll, one agent-scope acquire).
// Any other placement (B != 128: the groups' ids are not congruent mod 8; CU masks; partition modes) keeps the agent-scope
// release: with it forced on mixed placements the scan goes stale (tools/gpu/r05_call7.sh, as round 4 had found).
__device__ __forceinline__ void lp_arrive(unsigned* flag, bool fast) {
#ifdef LSTM_ABL_NOSYNC
  __syncthreads();
  return;
#endif
#if LSTM_XCHG == 1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // EVERY storing wave: its write-through payload stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  if (fast) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // EVERY storing wave: its payload stores are in the shared L2
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    __syncthreads();                    // every thread's rows of this step are written (and acknowledged by L2)
    if (threadIdx.x == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}
// XCD census of a group: before its FIRST arrive (an agent-scope release, which orders it) thread 0 of every block ORs
// 1 << XCC_ID into the group's mask word; after the first wait all 8 contributions are visible.  -> true iff the mask is
// exactly this block's own bit.  Called by thread 0 only.
__device__ __forceinline__ unsigned lp_xcc_bit() { return 1u << (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u); }
__device__ __forceinline__ void lp_census_vote(unsigned* mask_word) {
  __hip_atomic_fetch_or(mask_word, lp_xcc_bit(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool lp_same_xcd(unsigned* mask_word) {
  return __hip_atomic_load(mask_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lp_xcc_bit();
}
// 8 bf16 (16 bytes) of a partner's payload row
__device__ __forceinline__ rbf16x8 lp_load8(const __bf16* p) {
#if LSTM_XCHG == 1
  typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;
  unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<__bf16*>(p));
  const u64x2 v = {__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
  return __builtin_bit_cast(rbf16x8, v);
#else
  return *reinterpret_cast<const rbf16x8*>(p);
#endif
}
// the scan's fp32 outputs (gates, c, h / dgx: read by LATER kernels only).  LSTM_NT_OUT = 1: nontemporal stores, so that the
// lines do not sit dirty in L2 where the next step's release (an L2 write-back at agent scope) has to flush them
#ifndef LSTM_NT_OUT
#define LSTM_NT_OUT 0
#endif
__device__ __forceinline__ void lp_out(float* p, float v) {
#if LSTM_NT_OUT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void lp_store4(__bf16* p, unsigned long long v) {
#if LSTM_XCHG == 1
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  *reinterpret_cast<unsigned long long*>(p) = v;
#endif
}

// -DLP_TRACE (variant library `python __graft_entry__.py --variant lptrace LP_TRACE`, tools/dev/lstm_phases.py): thread 0 of
// every block stamps the 100 MHz wall clock at the phase boundaries of every time step of the FORWARD scan --
//   0 step begins (gx requested) | 1 partners' step counter seen, L1 invalidated, block released | 2 h_{t-1} tile in LDS
//   (block barrier) | 3 24 MFMAs done | 4 K halves folded through LDS | 5 gates, c, h computed, payload
//   staged | 6 payload stores issued | 7 release increment done (lp_arrive returned); fp32 outputs follow
// -- into a device array the tool reads back; XCC_ID of the block in slot 7 of step 0's row... see focr_lstm_trace_dump.
#ifdef LP_TRACE
#define LP_TRACE_STEPS 32
__device__ unsigned long long lp_trace_buf[256 * LP_TRACE_STEPS * 8];
__device__ unsigned lp_trace_xcc[256];
#define LP_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0 && step < LP_TRACE_STEPS && blockIdx.x < 256)                                           \
      lp_trace_buf[((size_t)blockIdx.x * LP_TRACE_STEPS + step) * 8 + (k)] = wall_clock64();                     \
  } while (0)
extern "C" int focr_lstm_trace_dump(unsigned long long* host_stamps, unsigned* host_xcc) {
  if (hipMemcpyFromSymbol(host_stamps, HIP_SYMBOL(lp_trace_buf), sizeof(lp_trace_buf)) != hipSuccess) return FOCR_EHIP;
  if (hipMemcpyFromSymbol(host_xcc, HIP_SYMBOL(lp_trace_xcc), sizeof(lp_trace_xcc)) != hipSuccess) return FOCR_EHIP;
  return FOCR_OK;
}
#else
#define LP_STAMP(k)
#endif

__global__ __launch_bounds__(512) void lstm_fwd_persist_bx3_kernel(
    const float* __restrict__ gx, const __bf16* __restrict__ whh2, const float* __restrict__ bhh,
    float* __restrict__ hseq, __bf16* __restrict__ hseq2, float* __restrict__ gates, float* __restrict__ cseq,
    unsigned* __restrict__ flags, int T, int B, int st_t, int st_b, int ngroups, int allow_fast, unsigned base) {
  constexpr int H = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lp_smem[];
  __bf16* Wh = reinterpret_cast<__bf16*>(lp_smem);              // [128][LP_WP]
  __bf16* Wl = Wh + 128 * LP_WP;
  float(*part)[32][33] = reinterpret_cast<float(*)[32][33]>(Wl + 128 * LP_WP);   // [4 gates][32][33]
  const int tid = threadIdx.x, wave = tid >> 6, g = wave & 3, kq = wave >> 2;
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int group = blockIdx.x % ngroups, jx = blockIdx.x / ngroups;
  const int dir = group & 1, j0 = jx * 32, b0 = (group >> 1) * 32;
  const long nh = (long)T * B * 2 * H;
  const long nw = (long)2 * 4 * H * H;
  unsigned* flag = flags + group;
  unsigned* err = flags + ngroups;
  __shared__ unsigned lp_bad, lp_fast;
  unsigned* const xmask = flags + ngroups + 1 + group;      // the group's XCD census word (see lp_census_vote)
  if (threadIdx.x == 0) {
    lp_bad = 0u;
    lp_fast = 0u;
    lp_census_vote(xmask);
  }
  bool fast = false;
  {   // 16-byte pieces of the 128 rows x 256 k slice: all sixteen requests of a thread in flight, then the LDS stores
    uint4 wv[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 512 * k, row = i >> 5, ch = i & 31, gg = row >> 5, u = row & 31;
      const __bf16* src = whh2 + ((size_t)dir * 4 * H + gg * H + j0 + u) * H + 8 * ch;
      wv[k][0] = *reinterpret_cast<const uint4*>(src);
      wv[k][1] = *reinterpret_cast<const uint4*>(src + nw);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 512 * k, row = i >> 5, ch = i & 31;
      *reinterpret_cast<uint4*>(Wh + row * LP_WP + 8 * ch) = wv[k][0];
      *reinterpret_cast<uint4*>(Wl + row * LP_WP + 8 * ch) = wv[k][1];
    }
  }
  // the two cell elements of this thread: (sequence ebl, unit eu), ebl = e >> 5 for e = tid, tid + 512
  float cprev[2] = {0.f, 0.f};
  const int eu = tid & 31;
  const float4 bias = make_float4(bhh[(size_t)dir * 4 * H + j0 + eu], bhh[(size_t)dir * 4 * H + H + j0 + eu],
                                  bhh[(size_t)dir * 4 * H + 2 * H + j0 + eu], bhh[(size_t)dir * 4 * H + 3 * H + j0 + eu]);
  __syncthreads();
#ifdef LP_TRACE
  if (threadIdx.x == 0 && blockIdx.x < 256) lp_trace_xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
#endif
  // Round 5 (phase stamps, profiles/r05_lstm_phases_*.txt: 2.5 of 7.6 us per step went into sixteen scattered 16-byte
  // loads of the partners' h rows per lane, 1.3 us into the W fragment reads interleaved with the MFMAs):
  //   * this wave's W_hh fragments -- rows g * 32 + li, columns 128 kq .. + 127, hi and lo: the same sixteen 16-byte
  //     pieces in EVERY step -- are read from the LDS image ONCE and stay in 64 VGPRs for the whole scan (the LDS copy only
  //     serves this transposition); the per-step matrix phase is 24 MFMAs and nothing else;
  //   * the LDS behind W is then free: h_{t-1} of the group's 32 sequences (hi + lo, 32 KB) is fetched by the whole block
  //     as whole 512-byte rows (4 coalesced 16-byte loads per thread instead of 16 row-scattered ones per lane, each
  //     row fetched once instead of once per gate wave) and the A fragments come from there.
  rbf16x8 wfh[8], wfl[8];
  {
    const __bf16* brow = Wh + (g * 32 + li) * LP_WP + 128 * kq + 8 * lh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      wfh[i] = *reinterpret_cast<const rbf16x8*>(brow + 16 * i);
      wfl[i] = *reinterpret_cast<const rbf16x8*>(brow + 128 * LP_WP + 16 * i);
    }
  }
  __syncthreads();                                   // every wave holds its fragments: the W image may be overwritten
  __bf16* Hs = Wh;                                   // [2 planes][32 sequences][LP_WP]
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    const int tp = dir == 0 ? t - 1 : t + 1;
    LP_STAMP(0);
    float gxv[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int eb = b0 + (tid >> 5) + 16 * e;
      const float* gp = gx + ((size_t)t * st_t + (size_t)min(eb, B - 1) * st_b) * 8 * H + dir * 4 * H + j0 + eu;
#pragma unroll
      for (int q = 0; q < 4; ++q) gxv[e][q] = gp[q * H];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (step > 0) {
      lp_wait(flag, base + 8u * (unsigned)step, err, &lp_bad);
      if (step == 1 && allow_fast) {                    // all 8 partners have voted: may the releases stay inside the XCD?
        if (threadIdx.x == 0) lp_fast = lp_same_xcd(xmask) ? 1u : 0u;
        __syncthreads();
        fast = lp_fast != 0u;
      }
      LP_STAMP(1);
      // the group's h_{t-1} tile: thread -> (plane, sequence, 16-byte chunk of the 512-byte row), 4 pieces each
      rbf16x8 piece[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = tid + 512 * k, pl = c >> 10, sq = (c >> 5) & 31, ch = c & 31;
        piece[k] = lp_load8(hseq2 + (size_t)pl * nh + ((size_t)tp * B + min(b0 + sq, B - 1)) * 2 * H + dir * H + 8 * ch);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = tid + 512 * k, pl = c >> 10, sq = (c >> 5) & 31, ch = c & 31;
        *reinterpret_cast<rbf16x8*>(Hs + (pl * 32 + sq) * LP_WP + 8 * ch) = piece[k];
      }
      __syncthreads();
      LP_STAMP(2);
      const __bf16* arow = Hs + li * LP_WP + 128 * kq + 8 * lh;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const rbf16x8 ah = *reinterpret_cast<const rbf16x8*>(arow + 16 * i);
        const rbf16x8 al = *reinterpret_cast<const rbf16x8*>(arow + 32 * LP_WP + 16 * i);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wfh[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wfl[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wfh[i], acc, 0, 0, 0);
      }
    }
#ifdef LP_TRACE
    asm volatile("s_nop 0" : "+v"(acc));          // (the accumulators are complete before the stamp)
    LP_STAMP(3);
#endif
    if (kq == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[g][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        part[g][bl][li] = acc[r] + part[g][bl][li];             // fixed order: low K half + high K half
      }
    }
    __syncthreads();
    LP_STAMP(4);
    // the partners only need the bf16 hi/lo copy of h: it goes out first and the step counter right behind it; the
    // fp32 outputs the BACKWARD pass reads (gates, c, h) are stored after the signal, off the critical path
    float o_ig[2], o_fg[2], o_gg[2], o_og[2], o_c[2], o_h[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ebl = (tid >> 5) + 16 * e, eb = b0 + ebl;
      o_ig[e] = sigmoidf_(part[0][ebl][eu] + gxv[e][0] + bias.x);
      o_fg[e] = sigmoidf_(part[1][ebl][eu] + gxv[e][1] + bias.y);
      o_gg[e] = tanhf(part[2][ebl][eu] + gxv[e][2] + bias.z);
      o_og[e] = sigmoidf_(part[3][ebl][eu] + gxv[e][3] + bias.w);
      o_c[e] = o_fg[e] * cprev[e] + o_ig[e] * o_gg[e];
      o_h[e] = o_og[e] * tanhf(o_c[e]);
      if (lp_bad) o_h[e] = __int_as_float(0x7fc00000);     // a wait timed out: poison, never a silently stale scan
      cprev[e] = o_c[e];
    }
    // payload: h as bf16 hi / lo, [plane][sequence][unit], staged in the (now idle) partial-sum tile and sent as one
    // 8-byte write-through store per thread
    __bf16* pay = reinterpret_cast<__bf16*>(&part[0][0][0]);
    __syncthreads();                               // every thread has read its partial sums
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ebl = (tid >> 5) + 16 * e;
      const __bf16 hh = (__bf16)o_h[e];
      pay[ebl * 32 + eu] = hh;
      pay[1024 + ebl * 32 + eu] = (__bf16)(o_h[e] - (float)hh);
    }
    __syncthreads();
    LP_STAMP(5);
    {
      const int pl = tid >> 8, sq = (tid >> 3) & 31, ch = tid & 7;
      if (b0 + sq < B)
        lp_store4(hseq2 + (size_t)pl * nh + ((size_t)t * B + b0 + sq) * 2 * H + dir * H + j0 + 4 * ch,
                  *reinterpret_cast<const unsigned long long*>(&pay[pl * 1024 + sq * 32 + 4 * ch]));
    }
    LP_STAMP(6);
    if (step + 1 < T) lp_arrive(flag, fast);    // (the barrier inside also protects `part` for the next step)
    LP_STAMP(7);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ebl = (tid >> 5) + 16 * e, eb = b0 + ebl;
      if (eb < B) {
        const size_t gb = (((size_t)t * B + eb) * 2 + dir) * 4 * H + j0 + eu;
        lp_out(&gates[gb], o_ig[e]);
        lp_out(&gates[gb + H], o_fg[e]);
        lp_out(&gates[gb + 2 * H], o_gg[e]);
        lp_out(&gates[gb + 3 * H], o_og[e]);
        lp_out(&cseq[(((size_t)t * B + eb) * 2 + dir) * H + j0 + eu], o_c[e]);
        lp_out(&hseq[((size_t)t * B + eb) * 2 * H + dir * H + j0 + eu], o_h[e]);
      }
    }
  }
}

__global__ __launch_bounds__(512) void lstm_bwd_persist_bx3_kernel(
    const float* __restrict__ dhseq, const __bf16* __restrict__ whhT2, const float* __restrict__ gates,
    const float* __restrict__ cseq, float* __restrict__ dgx, __bf16* __restrict__ dgx2, unsigned* __restrict__ flags,
    int T, int B, int st_t, int st_b, long ndg, int ngroups, int allow_fast, unsigned base) {
  constexpr int H = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char lp_smem[];
  __bf16* Wh = reinterpret_cast<__bf16*>(lp_smem);              // [32 units][LP_WTP]  (W_hh^T rows: n over 4H)
  __bf16* Wl = Wh + 32 * LP_WTP;
  float(*part)[32][33] = reinterpret_cast<float(*)[32][33]>(Wl + 32 * LP_WTP);   // [4][32][33]
  const int tid = threadIdx.x, wave = tid >> 6;
  const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int group = blockIdx.x % ngroups, jx = blockIdx.x / ngroups;
  const int dir = group & 1, j0 = jx * 32, b0 = (group >> 1) * 32;
  const long nwt = (long)2 * H * 4 * H;
  unsigned* flag = flags + group;
  unsigned* err = flags + ngroups;
  __shared__ unsigned lp_bad, lp_fast;
  unsigned* const xmask = flags + ngroups + 1 + group;      // the group's XCD census word (see lp_census_vote)
  if (threadIdx.x == 0) {
    lp_bad = 0u;
    lp_fast = 0u;
    lp_census_vote(xmask);
  }
  bool fast = false;
  {   // 16-byte pieces of 32 rows x 1024 n: all sixteen requests of a thread in flight, then the LDS stores
    uint4 wv[8][2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 512 * k, row = i >> 7, ch = i & 127;
      const __bf16* src = whhT2 + ((size_t)dir * H + j0 + row) * 4 * H + 8 * ch;
      wv[k][0] = *reinterpret_cast<const uint4*>(src);
      wv[k][1] = *reinterpret_cast<const uint4*>(src + nwt);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + 512 * k, row = i >> 7, ch = i & 127;
      *reinterpret_cast<uint4*>(Wh + row * LP_WTP + 8 * ch) = wv[k][0];
      *reinterpret_cast<uint4*>(Wl + row * LP_WTP + 8 * ch) = wv[k][1];
    }
  }
  float carry[2] = {0.f, 0.f};
  const int eu = tid & 31;
  __syncthreads();
  // as in the forward scan (round 5): this wave's W_hh^T fragments (rows = 32 units, its 128 gate columns, hi + lo) are
  // the same in every step -> 64 VGPRs for the whole scan; the LDS image behind them then holds the group's gate-gradient
  // tile of the previous step (32 sequences x 4H, hi + lo = 128 KB), fetched by the whole block as whole 2-KB rows
  rbf16x8 wfh[8], wfl[8];
  {
    const __bf16* brow = Wh + li * LP_WTP + 128 * wave + 8 * lh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      wfh[i] = *reinterpret_cast<const rbf16x8*>(brow + 16 * i);
      wfl[i] = *reinterpret_cast<const rbf16x8*>(brow + 32 * LP_WTP + 16 * i);
    }
  }
  __syncthreads();
  __bf16* Ds = Wh;                                   // [2 planes][32 sequences][LP_WTP]
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? T - 1 - step : step;
    const int tn = dir == 0 ? t + 1 : t - 1;
    const int tp = dir == 0 ? t - 1 : t + 1;
    const bool first = dir == 0 ? t == 0 : t == T - 1;
    float dh0[2], ig[2], fg[2], gg[2], og[2], c[2], cp[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int eb = min(b0 + (tid >> 5) + 16 * e, B - 1);
      dh0[e] = dhseq[((size_t)t * B + eb) * 2 * H + dir * H + j0 + eu];
      const size_t gb = (((size_t)t * B + eb) * 2 + dir) * 4 * H + j0 + eu;
      ig[e] = gates[gb], fg[e] = gates[gb + H], gg[e] = gates[gb + 2 * H], og[e] = gates[gb + 3 * H];
      c[e] = cseq[(((size_t)t * B + eb) * 2 + dir) * H + j0 + eu];
      cp[e] = first ? 0.f : cseq[(((size_t)tp * B + eb) * 2 + dir) * H + j0 + eu];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (step > 0) {
      lp_wait(flag, base + 8u * (unsigned)step, err, &lp_bad);
      if (step == 1 && allow_fast) {                    // all 8 partners have voted: may the releases stay inside the XCD?
        if (threadIdx.x == 0) lp_fast = lp_same_xcd(xmask) ? 1u : 0u;
        __syncthreads();
        fast = lp_fast != 0u;
      }
      // the group's gate-gradient tile of step t_next: thread -> (plane, sequence, 16-byte chunk of the 2-KB row), 16 pieces
      // each in two batches of 8 (register budget)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        rbf16x8 piece[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = tid + 512 * (8 * half + k), pl = c >> 12, sq = (c >> 7) & 31, ch = c & 127;
          piece[k] = lp_load8(dgx2 + (size_t)pl * ndg +
                              ((size_t)tn * st_t + (size_t)min(b0 + sq, B - 1) * st_b) * 8 * H + dir * 4 * H + 8 * ch);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = tid + 512 * (8 * half + k), pl = c >> 12, sq = (c >> 7) & 31, ch = c & 127;
          *reinterpret_cast<rbf16x8*>(Ds + (pl * 32 + sq) * LP_WTP + 8 * ch) = piece[k];
        }
      }
      __syncthreads();
      // wave w contracts n in [128 w, 128 w + 128) of the 4H gate-gradient row
      const __bf16* arow = Ds + li * LP_WTP + 128 * wave + 8 * lh;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const rbf16x8 ah = *reinterpret_cast<const rbf16x8*>(arow + 16 * i);
        const rbf16x8 al = *reinterpret_cast<const rbf16x8*>(arow + 32 * LP_WTP + 16 * i);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wfh[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wfl[i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wfh[i], acc, 0, 0, 0);
      }
    }
    // fold the 8 partial tiles in a fixed order: (w + (w + 4)) per slot, then slots 0..3 in the epilogue
    if (wave >= 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave - 4][(r & 3) + 8 * (r >> 2) + 4 * lh][li] = acc[r];
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int bl = (r & 3) + 8 * (r >> 2) + 4 * lh;
        part[wave][bl][li] = acc[r] + part[wave][bl][li];
      }
    }
    __syncthreads();
    float d4[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ebl = (tid >> 5) + 16 * e, eb = b0 + ebl;
      const float dh = dh0[e] + (((part[0][ebl][eu] + part[1][ebl][eu]) + part[2][ebl][eu]) + part[3][ebl][eu]);
      const float tc = tanhf(c[e]);
      const float dc = dh * og[e] * (1.f - tc * tc) + carry[e];
      carry[e] = dc * fg[e];
      d4[e][0] = dc * gg[e] * ig[e] * (1.f - ig[e]);
      d4[e][1] = dc * cp[e] * fg[e] * (1.f - fg[e]);
      d4[e][2] = dc * ig[e] * (1.f - gg[e] * gg[e]);
      d4[e][3] = dh * tc * og[e] * (1.f - og[e]);
      if (lp_bad) d4[e][0] = d4[e][1] = d4[e][2] = d4[e][3] = __int_as_float(0x7fc00000);
    }
    // payload: the gate gradients as bf16 hi / lo, [plane][sequence][gate][unit] (16 KB), staged in the partial-sum
    // tile and sent as four 8-byte write-through stores per thread
    __bf16* pay = reinterpret_cast<__bf16*>(&part[0][0][0]);
    __syncthreads();                               // every thread has read its partial sums
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int ebl = (tid >> 5) + 16 * e;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const __bf16 hh = (__bf16)d4[e][q];
        pay[(ebl * 4 + q) * 32 + eu] = hh;
        pay[4096 + (ebl * 4 + q) * 32 + eu] = (__bf16)(d4[e][q] - (float)hh);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = tid + 512 * k, pl = c >> 10, sq = (c >> 5) & 31, q = (c >> 3) & 3, ch = c & 7;
      if (b0 + sq < B)
        lp_store4(dgx2 + (size_t)pl * ndg + ((size_t)t * st_t + (size_t)(b0 + sq) * st_b) * 8 * H + dir * 4 * H + q * H + j0 +
                      4 * ch,
                  *reinterpret_cast<const unsigned long long*>(&pay[pl * 4096 + (sq * 4 + q) * 32 + 4 * ch]));
    }
    if (step + 1 < T) lp_arrive(flag, fast);    // partners read the bf16 copies only; the fp32 result follows
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int eb = b0 + (tid >> 5) + 16 * e;
      if (eb < B) {
        const size_t ob = ((size_t)t * st_t + (size_t)eb * st_b) * 8 * H + dir * 4 * H + j0 + eu;
#pragma unroll
        for (int q = 0; q < 4; ++q) lp_out(&dgx[ob + q * H], d4[e][q]);
      }
    }
  }
}

#define LP_FWD_LDS (2 * 128 * LP_WP * 2 + 4 * 32 * 33 * 4)       // 152064 B
#define LP_BWD_LDS (2 * 32 * LP_WTP * 2 + 4 * 32 * 33 * 4)       // 148992 B
#define LP_FLAG_BYTES 1024                                      // step counters of the groups + error word
// The persistent scan synchronises its 8 * ngroups blocks through global step counters under an ordinary launch: it is
// only correct when ALL of them are resident at once.  Ask the runtime how many blocks of each kernel fit on this device
// (CU count under the current partition mode / CU mask x blocks per CU at ~150 KB LDS) and fall back to the per-step
// launches when the grid does not fit; a query that fails also means "not usable".
static int lp_resident_blocks(const void* kern, size_t lds) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return per_cu * cus;
}
static bool lp_usable(int B, int H) {
  if (!(H == 256 && 8 * cdiv(B, 32) * 2 <= 256 && focr_get_tuning(FOCR_TUNE_LSTM_PERSISTENT) != 0 &&
        2 * (cdiv(B, 32) * 2) + 1 <= LP_FLAG_BYTES / 4))
    return false;
  static std::atomic<int> resident_dev[64];          // min over the two kernels + 1 (0 = not asked yet), per device
  std::atomic<int>& slot = resident_dev[focr_cur_device()];
  int resident = slot.load(std::memory_order_relaxed) - 1;
  if (resident < 0) {
    const int f = lp_resident_blocks((const void*)lstm_fwd_persist_bx3_kernel, LP_FWD_LDS);
    const int b = lp_resident_blocks((const void*)lstm_bwd_persist_bx3_kernel, LP_BWD_LDS);
    resident = f < b ? f : b;
    slot.store(resident + 1, std::memory_order_relaxed);
  }
  return 8 * cdiv(B, 32) * 2 <= resident;
}

// The XCD-local release (lp_arrive `fast`) rests on two facts of THIS chip that the HIP memory model does not promise: plain
// stores of blocks on one XCD are visible to each other in that XCD's L2 once acknowledged, and hardware register 20 read
// through s_getreg is XCC_ID.  Both were verified on gfx950 in SPX mode (256 CUs, tools/gpu/r05_call7.sh / call8.sh); on
// any other device -- another architecture, or a partitioned gfx950 whose CU count differs -- the agent-scope release is
// used in every step.  Cached per device.
static bool lp_fast_release_verified() {
  static std::atomic<int> verdict[64];               // 0 = not asked, 1 = no, 2 = yes
  std::atomic<int>& slot = verdict[focr_cur_device()];
  int v = slot.load(std::memory_order_relaxed);
  if (v == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    v = 1;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
        strncmp(prop.gcnArchName, "gfx950", 6) == 0 && prop.multiProcessorCount == 256)
      v = 2;
    slot.store(v, std::memory_order_relaxed);
  }
  return v == 2;
}

// ws (forward): bf16 elements: 2*2*4H*H (whh hi/lo) + 2*T*B*2H (h hi/lo)          -> bytes = 2 * that
// ws (backward): bf16 elements: 2*2*H*4H (whh^T hi/lo) + 2*rows*8H (dgx hi/lo)
extern "C" long focr_lstm_ws_bytes(int T, int B, int H, int backward) {
  long w = (long)2 * 2 * 4 * H * H;
  long s = backward ? (long)2 * T * B * 8 * H : (long)2 * T * B * 2 * H;
  return 2 * (w + s) + LP_FLAG_BYTES;          // + the step counters of the persistent kernels (at the end)
}

int focr_lstm_split_weights(const float* whh, void* out, int H, int backward, hipStream_t stream) {
  const long nw = (long)2 * 4 * H * H;
  if (backward) hipLaunchKernelGGL(split_bf16_kernel, dim3(512), 256, 0, stream, whh, reinterpret_cast<__bf16*>(out), nw, 4 * H, H, 1);
  else hipLaunchKernelGGL(split_bf16_kernel, dim3(512), 256, 0, stream, whh, reinterpret_cast<__bf16*>(out), nw, 1, 1, 0);
  return 0;
}
// wsplit (optional): the hi / lo split of W_hh prepared once by focr_lstm_prepare_weights (frozen recognizer: the same
// every step) -- the per-call split launch is then skipped
// pflags (optional, with `base`): a caller-owned, never re-zeroed block of LP_FLAG_BYTES whose step counters simply keep
// counting -- every call adds 8 (T - 1) to each of its groups' words, the caller passes the sum so far as `base` (targets
// are compared modulo 2^32) -- instead of the workspace tail + a memset launch in front of every scan
int focr_lstm_fwd_bx3(const float* gx, const float* whh, const float* bhh, float* hseq, float* gates, float* cseq,
                      void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H, int st_t, int st_b,
                      hipStream_t stream) {
  __bf16* whh2 = reinterpret_cast<__bf16*>(ws);
  long nw = (long)2 * 4 * H * H;
  __bf16* hseq2 = whh2 + 2 * nw;
  if (wsplit) whh2 = const_cast<__bf16*>(reinterpret_cast<const __bf16*>(wsplit));
  else hipLaunchKernelGGL(split_bf16_kernel, dim3(512), 256, 0, stream, whh, whh2, nw, 1, 1, 0);
  if (lp_usable(B, H)) {
    static focr_dev_flags attr;
    if (focr_dev_first(attr)) {
      (void)hipFuncSetAttribute((const void*)lstm_fwd_persist_bx3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LP_FWD_LDS);
      focr_dev_mark(attr);
    }
    unsigned* flags = reinterpret_cast<unsigned*>(pflags);
    if (!flags) {
      flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + focr_lstm_ws_bytes(T, B, H, 0) - LP_FLAG_BYTES);
      (void)hipMemsetAsync(flags, 0, LP_FLAG_BYTES, stream);
      base = 0u;
    }
    const int ngroups = cdiv(B, 32) * 2;
    hipLaunchKernelGGL(lstm_fwd_persist_bx3_kernel, dim3(8 * ngroups), 512, LP_FWD_LDS, stream, gx,
                       (const __bf16*)whh2, bhh, hseq, hseq2, gates, cseq, flags, T, B, st_t, st_b, ngroups,
                       focr_get_tuning(FOCR_TUNE_LSTM_PERSISTENT) != 2 && lp_fast_release_verified(), base);
    return 0;
  }
  dim3 grid(H / 32, (B + 31) / 32, 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_fwd_step_bx3_kernel, grid, 1024, 0, stream, gx, (const __bf16*)whh2, bhh, hseq, hseq2, gates,
                       cseq, s, T, B, H, st_t, st_b);
  return 0;
}
int focr_lstm_bwd_bx3(const float* dhseq, const float* whh, const float* gates, const float* cseq, float* dgx,
                      float* dc_carry, void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H,
                      int st_t, int st_b, hipStream_t stream) {
  __bf16* whhT2 = reinterpret_cast<__bf16*>(ws);
  long nw = (long)2 * 4 * H * H;
  __bf16* dgx2 = whhT2 + 2 * nw;
  long ndg = (long)T * B * 8 * H;
  // whh [2][4H][H] -> whhT [2][H][4H] (hi plane then lo plane)
  if (wsplit) whhT2 = const_cast<__bf16*>(reinterpret_cast<const __bf16*>(wsplit));
  else hipLaunchKernelGGL(split_bf16_kernel, dim3(512), 256, 0, stream, whh, whhT2, nw, 4 * H, H, 1);
  if (lp_usable(B, H)) {
    static focr_dev_flags attr;
    if (focr_dev_first(attr)) {
      (void)hipFuncSetAttribute((const void*)lstm_bwd_persist_bx3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LP_BWD_LDS);
      focr_dev_mark(attr);
    }
    unsigned* flags = reinterpret_cast<unsigned*>(pflags);
    if (!flags) {
      flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + focr_lstm_ws_bytes(T, B, H, 1) - LP_FLAG_BYTES);
      (void)hipMemsetAsync(flags, 0, LP_FLAG_BYTES, stream);
      base = 0u;
    }
    const int ngroups = cdiv(B, 32) * 2;
    hipLaunchKernelGGL(lstm_bwd_persist_bx3_kernel, dim3(8 * ngroups), 512, LP_BWD_LDS, stream, dhseq,
                       (const __bf16*)whhT2, gates, cseq, dgx, dgx2, flags, T, B, st_t, st_b, ndg, ngroups,
                       focr_get_tuning(FOCR_TUNE_LSTM_PERSISTENT) != 2 && lp_fast_release_verified(), base);
    return 0;
  }
  dim3 grid(H / 32, (B + 31) / 32, 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_bwd_step_bx3_kernel, grid, 1024, 0, stream, dhseq, (const __bf16*)whhT2, gates, cseq, dgx,
                       dgx2, dc_carry, s, T, B, H, st_t, st_b, ndg);
  return 0;
}

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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

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
extern "C" int focr_lstm_bidir_fwd(const float* gx, const float* whh, const float* bhh, float* hseq,
                                   float* gates, float* cseq, int T, int B, int H, int st_t, int st_b,
                                   hipStream_t stream) {
  FOCR_CHECK_ARG(gx && whh && bhh && hseq && gates && cseq, "null pointer");
  FOCR_CHECK_ARG(T > 0 && B > 0 && H % 32 == 0, "need H % 32 == 0");
  dim3 grid(H / 32, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, 256, 0, stream, gx, whh, bhh, hseq, gates, cseq, s, T, B, H,
                       st_t, st_b);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// dc_carry: 2*B*H floats of workspace.  dgx is fully overwritten.
extern "C" int focr_lstm_bidir_bwd(const float* dhseq, const float* whh, const float* gates,
                                   const float* cseq, float* dgx, float* dc_carry, int T, int B, int H,
                                   int st_t, int st_b, hipStream_t stream) {
  FOCR_CHECK_ARG(dhseq && whh && gates && cseq && dgx && dc_carry, "null pointer");
  FOCR_CHECK_ARG(T > 0 && B > 0 && H % 32 == 0, "need H % 32 == 0");
  dim3 grid(H / 32, cdiv(B, 32), 2);
  for (int s = 0; s < T; ++s)
    hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, 256, 0, stream, dhseq, whh, gates, cseq, dgx, dc_carry, s, T,
                       B, H, st_t, st_b);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// log_softmax + CTC negative log-likelihood + gradient w.r.t. the logits, one wave per sample.
//
// The reference has no CTC loss: the measured step composes the
// reference CRNN with torch.nn.functional.ctc_loss(log_softmax(logits), blank=0,
// reduction='mean', zero_infinity=True).  This kernel restates that composition
// (Graves et al. 2006 alpha/beta recursion in log space); label codec per
// utils/utils_crnn.py:21-53 (blank 0, '0-9a-z' -> 1..36).
//
// Wavefront scan: lane s owns extended-label position s (S = 2L+1 <= 64); the T time steps are
// sequential; alpha_{t-1}(s-1), alpha_{t-1}(s-2) come from LDS.  The gradient uses
//   d nll / d logit[t][c] = softmax[t][c] - sum_{s: ext(s)=c} exp(alpha_t(s)+beta_t(s)-lp[t][c]-ll)
// and is already scaled by 1/(L*B) ('mean' reduction); infeasible samples give loss 0, grad 0.
// Limits: T <= 64, C <= 64 (checked on the host) and label length L <= 31 (S = 2L+1 <= 64 lanes): lengths live on the
// device, so the kernel treats a longer label as infeasible (loss 0, gradient 0 -- never an out-of-range LDS access);
// the Python wrapper (loss/ctc_focus_loss.py encode) rejects such labels while they are still host data.
// The batch loss is reduced in a fixed order by ctc_reduce_kernel: bit-identical run to run (no atomics).
#include "focr_common.h"

#define CTC_TMAX 64
#define CTC_NEG (-1e30f)

__device__ __forceinline__ float lse2(float a, float b) {
  float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

__global__ __launch_bounds__(64) void ctc_kernel(const float* __restrict__ logits,   // [T,B,C]
                                                 const int* __restrict__ targets,    // [sum L]
                                                 const int* __restrict__ tlen,       // [B]
                                                 const int* __restrict__ toff,       // [B]
                                                 float* __restrict__ loss_sum,       // [1] (written by ctc_reduce_kernel)
                                                 float* __restrict__ nll_out,        // [B] raw nll
                                                 float* __restrict__ grad,           // [T,B,C]
                                                 int T, int B, int C) {
  __shared__ float lp[CTC_TMAX][64];
  __shared__ float alpha[CTC_TMAX][64];
  __shared__ float bet[2][64];
  __shared__ float occ[64];
  const int b = blockIdx.x, s = threadIdx.x;
  const int Lraw = tlen[b];
  const bool too_long = Lraw > 31 || Lraw < 0;      // would not fit the 64-lane lattice: handled as infeasible
  const int L = too_long ? 0 : Lraw;
  const int S = 2 * L + 1;
  const int off = toff[b];
  // extended label of this lane
  int ext = 0;
  if (s < S && (s & 1)) ext = targets[off + (s >> 1)];
  bool skip_ok = false;   // transition s-2 -> s allowed
  if (s < S && (s & 1) && s >= 2) skip_ok = targets[off + (s >> 1)] != targets[off + (s >> 1) - 1];
  bool skip_fw = false;   // transition s -> s+2 allowed (for beta)
  if (s + 2 < S && (s & 1)) skip_fw = targets[off + (s >> 1) + 1] != targets[off + (s >> 1)];

  // ---- log-softmax of every time step (lanes = classes) ----
  for (int t = 0; t < T; ++t) {
    float v = s < C ? logits[((size_t)t * B + b) * C + s] : CTC_NEG;
    float mx = wave_max(v);
    float e = s < C ? expf(v - mx) : 0.f;
    float sum = wave_sum(e);
    lp[t][s] = s < C ? v - mx - logf(sum) : CTC_NEG;
  }
  __syncthreads();
  // ---- alpha ----
  {
    float a = CTC_NEG;
    if (s == 0) a = lp[0][0];
    if (s == 1 && S > 1) a = lp[0][ext];
    alpha[0][s] = a;
  }
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    float a = CTC_NEG;
    if (s < S) {
      a = alpha[t - 1][s];
      if (s >= 1) a = lse2(a, alpha[t - 1][s - 1]);
      if (skip_ok) a = lse2(a, alpha[t - 1][s - 2]);
      a += lp[t][ext];
      if (a < CTC_NEG) a = CTC_NEG;
    }
    alpha[t][s] = a;
    __syncthreads();
  }
  float ll = alpha[T - 1][S - 1];
  if (S > 1) ll = lse2(ll, alpha[T - 1][S - 2]);
  const bool feasible = ll > -1e29f && !too_long;
  const float norm = 1.f / ((float)(L > 0 ? L : 1) * (float)B);
  if (s == 0) nll_out[b] = feasible ? -ll : 0.f;      // summed in fixed order by ctc_reduce_kernel
  // ---- beta + gradient ----
  int cur = 0;
  {
    float bt = CTC_NEG;
    if (s == S - 1) bt = lp[T - 1][0];
    if (S > 1 && s == S - 2) bt = lp[T - 1][ext];
    bet[0][s] = bt;
  }
  for (int t = T - 1; t >= 0; --t) {
    __syncthreads();
    if (t < T - 1) {
      float bt = CTC_NEG;
      if (s < S) {
        bt = bet[cur][s];
        if (s + 1 < S) bt = lse2(bt, bet[cur][s + 1]);
        if (skip_fw) bt = lse2(bt, bet[cur][s + 2]);
        bt += lp[t][ext];
        if (bt < CTC_NEG) bt = CTC_NEG;
      }
      bet[cur ^ 1][s] = bt;
      cur ^= 1;
    }
    occ[s] = 0.f;
    __syncthreads();
    if (feasible && s < S) {
      float ab = alpha[t][s] + bet[cur][s] - lp[t][ext] - ll;
      if (ab > -80.f) atomicAdd(&occ[ext], expf(ab));
    }
    __syncthreads();
    if (s < C) {
      float gval = feasible ? (expf(lp[t][s]) - occ[s]) * norm : 0.f;
      grad[((size_t)t * B + b) * C + s] = gval;
    }
  }
}

// loss = sum_b nll[b] / (max(L_b, 1) * B): lane i adds samples i, i + 64, ... in order, then a fixed shuffle tree
__global__ __launch_bounds__(64) void ctc_reduce_kernel(const float* __restrict__ nll, const int* __restrict__ tlen,
                                                        float* __restrict__ loss, int B) {
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const int L = tlen[b];
    acc += nll[b] / ((float)((L > 0 && L <= 31) ? L : 1) * (float)B);
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = acc;
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ sc, float* __restrict__ y,
                                 long n) {
  const float k = sc[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = k * x[i];
}

extern "C" int focr_ctc_fwd(const float* logits, const int* targets, const int* target_lengths,
                            const int* target_offsets, float* loss, float* nll, float* grad_logits, int T,
                            int B, int C, hipStream_t stream) {
  FOCR_CHECK_ARG(logits && targets && target_lengths && target_offsets && loss && nll && grad_logits, "null pointer");
  FOCR_CHECK_ARG(T > 0 && T <= CTC_TMAX && C > 1 && C <= 64 && B > 0, "need T <= 64, C <= 64");
  hipLaunchKernelGGL(ctc_kernel, dim3(B), 64, 0, stream, logits, targets, target_lengths, target_offsets, loss, nll,
                     grad_logits, T, B, C);
  hipLaunchKernelGGL(ctc_reduce_kernel, dim3(1), 64, 0, stream, (const float*)nll, target_lengths, loss, B);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// y = s[0] * x  with the scalar living on the device (no host sync in backward chains)
extern "C" int focr_scale_dev(const float* x, const float* s, float* y, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(x && s && y && n > 0, "bad argument");
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(scale_dev_kernel, dim3((int)g), 256, 0, stream, x, s, y, n);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

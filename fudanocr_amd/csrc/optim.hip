// Fused optimiser tail of the training step (interfaces/super_resolution.py:83-84,
// interfaces/base.py:194-198): torch.nn.utils.clip_grad_norm_(params, 0.25) followed by
// Adam(lr, betas=(beta1, 0.999), eps 1e-8, bias-corrected) on ONE flat fp32 buffer (the same
// buffer the data-parallel all-reduce runs on).  The clip coefficient is computed on the device
// from the squared-norm accumulator: no host synchronisation anywhere in the step.
// HBM-bound: reads g,p,m,v, writes p,m,v = 28 B per parameter.
#include "focr_common.h"

#define SUMSQ_BLOCKS 1024

// part[1 + block] = this block's sum of g^2.  No atomics: the squared norm must be bit-identical on every
// data-parallel rank (identical all-reduced gradients -> identical clip factor -> replicas never drift apart).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, float* __restrict__ part, long n4,
                                                    long n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// part[0] = gscale^2 * sum of the block partials, folded in a fixed order by one block
__global__ __launch_bounds__(256) void sumsq_fold_kernel(float* __restrict__ part, int nblk, float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) acc += part[1 + i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[0] = ((red[0] + red[1]) + (red[2] + red[3])) * gscale * gscale;
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ sumsq, long n, float lr,
                                                        float b1, float b2, float eps, float c1, float c2s,
                                                        float max_norm, float gscale) {
  // clip_grad_norm_: coef = max_norm / (norm + 1e-6), applied only when < 1
  float norm = sqrtf(sumsq[0]);
  float coef = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
  const float k = coef * gscale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * k;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    // torch: denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= lr/(1-b1^t) * m/denom
    p[i] -= (lr / c1) * mi / (sqrtf(vi) / c2s + eps);
  }
}

extern "C" long focr_grad_sumsq_ws_floats(void) { return 1 + SUMSQ_BLOCKS; }

// sumsq: focr_grad_sumsq_ws_floats() floats; sumsq[0] is overwritten with the squared (averaged) gradient norm.
extern "C" int focr_grad_sumsq(const float* g, float* sumsq, long n, float gscale, hipStream_t stream) {
  FOCR_CHECK_ARG(g && sumsq && n > 0, "bad argument");
  long n4 = n / 4;
  long gsz = (n4 + 255) / 256;
  if (gsz > SUMSQ_BLOCKS) gsz = SUMSQ_BLOCKS;
  if (gsz < 1) gsz = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)gsz), 256, 0, stream, g, sumsq, n4, n);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), 256, 0, stream, sumsq, (int)gsz, gscale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ---- step state (focr_core.hip): advance = epoch + 1, t + 1, bias corrections of the new t
__global__ void step_advance_kernel(unsigned long long* __restrict__ st, double beta1, double beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    st[0] += 1ull;
    const long long t = (long long)st[1] + 1;
    st[1] = (unsigned long long)t;
    float* f = reinterpret_cast<float*>(st + 2);
    f[0] = (float)(1.0 - pow(beta1, (double)t));
    f[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
  }
}
extern "C" int focr_step_state_bytes(void) { return 64; }
extern "C" int focr_step_advance(void* state, double beta1, double beta2, hipStream_t stream) {
  FOCR_CHECK_ARG(state && (reinterpret_cast<size_t>(state) & 7) == 0, "needs an 8-byte aligned 64-byte state block");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), 64, 0, stream, reinterpret_cast<unsigned long long*>(state), beta1, beta2);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
__global__ __launch_bounds__(256) void clip_adam_state_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              const float* __restrict__ sumsq, long n, float lr,
                                                              float b1, float b2, float eps,
                                                              const float* __restrict__ corr, float max_norm,
                                                              float gscale) {
  const float c1 = corr[0], c2s = corr[1];
  float norm = sqrtf(sumsq[0]);
  float coef = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
  const float k = coef * gscale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * k;
    float mi = b1 * m[i] + (1.f - b1) * gi;
    float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / c1) * mi / (sqrtf(vi) / c2s + eps);
  }
}
// focr_clip_adam with the step count / bias corrections read from the device-resident step state (after focr_step_advance)
extern "C" int focr_clip_adam_state(float* p, const float* g, float* m, float* v, const float* sumsq, long n, float lr,
                                    float beta1, float beta2, float eps, const void* state, float max_norm, float gscale,
                                    hipStream_t stream) {
  FOCR_CHECK_ARG(p && g && m && v && sumsq && n > 0 && state, "bad argument");
  long gsz = (n + 255) / 256;
  if (gsz > 2048) gsz = 2048;
  hipLaunchKernelGGL(clip_adam_state_kernel, dim3((int)gsz), 256, 0, stream, p, g, m, v, sumsq, n, lr, beta1, beta2, eps,
                     reinterpret_cast<const float*>(reinterpret_cast<const char*>(state) + 16), max_norm, gscale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// step: 1-based Adam step count.  max_norm <= 0 disables clipping.
extern "C" int focr_clip_adam(float* p, const float* g, float* m, float* v, const float* sumsq, long n,
                              float lr, float beta1, float beta2, float eps, int step, float max_norm,
                              float gscale, hipStream_t stream) {
  FOCR_CHECK_ARG(p && g && m && v && sumsq && n > 0 && step >= 1, "bad argument");
  float c1 = 1.f - powf(beta1, (float)step);
  float c2s = sqrtf(1.f - powf(beta2, (float)step));
  long gsz = (n + 255) / 256;
  if (gsz > 2048) gsz = 2048;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((int)gsz), 256, 0, stream, p, g, m, v, sumsq, n, lr, beta1, beta2, eps,
                     c1, c2s, max_norm, gscale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// zero_grad of the flat gradient buffer (optimizer.zero_grad(), interfaces/super_resolution.py:82): float4 grid-stride
// stores.  torch's fill kernel took 61 us for the 12.8 MB buffer at the head of every step (profiles/r04_step_sequence.txt).
__global__ __launch_bounds__(256) void zero_kernel(float4* __restrict__ p, long n4, float* __restrict__ tail, int ntail) {
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
extern "C" int focr_zero(float* p, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(p && n > 0 && (reinterpret_cast<size_t>(p) & 15) == 0, "needs a 16-byte aligned buffer");
  const long n4 = n / 4;
  long gsz = (n4 + 255) / 256;
  if (gsz > 2048) gsz = 2048;
  if (gsz < 1) gsz = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((int)gsz), 256, 0, stream, reinterpret_cast<float4*>(p), n4, p + 4 * n4, (int)(n - 4 * n4));
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

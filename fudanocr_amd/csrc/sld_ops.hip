// Operators of the stroke-level-decomposition transformer recognizer (BASELINE configs[4]; reference
// /root/reference/stroke-level-decomposition/model/transformer.py, train.py) that the SR path does not have:
//   * generic-width LayerNorm (custom: unbiased std, eps on std; transformer.py:241-251, D = 1024),
//   * relu(a + b) of the ResNet BasicBlock (transformer.py:66-75),
//   * embedding lookup * sqrt(d) (transformer.py:269-277),
//   * attention for FEW queries and long heads (4 heads x d_k 256, <= 30 queries, 30 / 256 keys; optional causal mask
//     and dropout on the probabilities; transformer.py:184-238) -- one workgroup per (batch, head), fp32 FMA: the
//     decoder is 4 % of the model's flops, the 3x3 convolutions of the encoder run on the halo kernel,
//   * ragged row gather of the predictions (transformer.py:362-370), cross-entropy (train.py:41,70),
//   * Adadelta (train.py:36-38) fused over the flat parameter buffer.
#include "focr_common.h"

// ------------------------------------------------------------------------------------------------- LayerNorm, any D
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ln_any_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                         const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ y, float* __restrict__ save_mean,
                                                         float* __restrict__ save_rinv, int D, float eps) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) s += x[row * D + i] + (res ? res[row * D + i] : 0.f);
  const float mean = block_sum256(s, red) / D;
  float q = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float u = x[row * D + i] + (res ? res[row * D + i] : 0.f) - mean;
    q += u * u;
  }
  const float sd = sqrtf(block_sum256(q, red) / (D - 1));
  const float rinv = 1.f / (sd + eps);
  for (int i = threadIdx.x; i < D; i += 256) {
    const float u = x[row * D + i] + (res ? res[row * D + i] : 0.f) - mean;
    y[row * D + i] = a[i] * u * rinv + b[i];
  }
  if (threadIdx.x == 0) {
    save_mean[row] = mean;
    save_rinv[row] = rinv;
  }
}

// dx (also the residual's gradient); da / db accumulated with atomics (few hundred rows)
__global__ __launch_bounds__(256) void ln_any_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const float* __restrict__ res, const float* __restrict__ a,
                                                         const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_rinv, float* __restrict__ dx,
                                                         float* __restrict__ da, float* __restrict__ db, int D,
                                                         float eps) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float mean = save_mean[row], rinv = save_rinv[row], sd = 1.f / rinv - eps;
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float u = x[row * D + i] + (res ? res[row * D + i] : 0.f) - mean;
    const float dn = dy[row * D + i] * a[i];
    s1 += dn;
    s2 += dn * u;
  }
  const float s_dn = block_sum256(s1, red);
  const float s_dnu = block_sum256(s2, red);
  const float k = rinv * rinv * s_dnu / ((D - 1) * sd), mdn = rinv * s_dn / D;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float u = x[row * D + i] + (res ? res[row * D + i] : 0.f) - mean;
    const float g = dy[row * D + i];
    dx[row * D + i] = g * a[i] * rinv - k * u - mdn;
    atomicAdd(&da[i], g * u * rinv);
    atomicAdd(&db[i], g);
  }
}

int focr_ln_any_fwd(const float* x, const float* res, const float* a, const float* b, float* y, float* save_mean,
                    float* save_rinv, long rows, int D, float eps, hipStream_t stream) {
  hipLaunchKernelGGL(ln_any_fwd_kernel, dim3((int)rows), 256, 0, stream, x, res, a, b, y, save_mean, save_rinv, D, eps);
  return 1;
}
int focr_ln_any_bwd(const float* dy, const float* x, const float* res, const float* a, const float* save_mean,
                    const float* save_rinv, float* dx, float* da, float* db, long rows, int D, float eps,
                    hipStream_t stream) {
  hipLaunchKernelGGL(ln_any_bwd_kernel, dim3((int)rows), 256, 0, stream, dy, x, res, a, save_mean, save_rinv, dx, da,
                     db, D, eps);
  return 1;
}

// ------------------------------------------------------------------------------------------------- relu(a + b)
__global__ __launch_bounds__(256) void add_relu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(fmaxf(u.x + v.x, 0.f), fmaxf(u.y + v.y, 0.f), fmaxf(u.z + v.z, 0.f),
                                                  fmaxf(u.w + v.w, 0.f));
  }
}
extern "C" int focr_add_relu_fwd(const float* a, const float* b, float* y, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && y && n > 0 && n % 4 == 0, "need n % 4 == 0");
  long g = (n / 4 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(add_relu_fwd_kernel, dim3((int)g), 256, 0, stream, a, b, y, n / 4);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// backward = focr_relu_bwd(dy, y, g): the same gradient goes to both addends

// ------------------------------------------------------------------------------------------------- embedding
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const long long* __restrict__ idx,
                                                            const float* __restrict__ table, float* __restrict__ y,
                                                            long rows, int D, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * D; i += (long)gridDim.x * 256) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    y[i] = table[idx[r] * D + c] * scale;
  }
}
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const long long* __restrict__ idx,
                                                            const float* __restrict__ dy, float* __restrict__ dtable,
                                                            long rows, int D, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * D; i += (long)gridDim.x * 256) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    atomicAdd(&dtable[idx[r] * D + c], dy[i] * scale);
  }
}
extern "C" int focr_embedding_fwd(const long long* idx, const float* table, float* y, long rows, int D, float scale,
                                  hipStream_t stream) {
  FOCR_CHECK_ARG(idx && table && y && rows > 0 && D > 0, "bad argument");
  long g = (rows * D + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3((int)g), 256, 0, stream, idx, table, y, rows, D, scale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// dtable[V][D] is ACCUMULATED into (caller zeroes it, or passes a slice of the zeroed flat gradient buffer)
extern "C" int focr_embedding_bwd(const long long* idx, const float* dy, float* dtable, long rows, int D, float scale,
                                  hipStream_t stream) {
  FOCR_CHECK_ARG(idx && dy && dtable && rows > 0 && D > 0, "bad argument");
  long g = (rows * D + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(embedding_bwd_kernel, dim3((int)g), 256, 0, stream, idx, dy, dtable, rows, D, scale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------- ragged gather
// out[r] = in[idx[r]] (rows of D floats); mode 1: scatter (in[idx[r]] = out[r], the gather's backward; the caller
// zeroes `in` first -- the indices of the prediction gather are unique)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const long long* __restrict__ idx,
                                                          float* __restrict__ dst, long rows, int D, int scatter) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * D; i += (long)gridDim.x * 256) {
    const long r = i / D;
    const int c = (int)(i - r * D);
    if (scatter) dst[idx[r] * D + c] = src[i];
    else dst[i] = src[idx[r] * D + c];
  }
}
extern "C" int focr_gather_rows(const float* src, const long long* idx, float* dst, long rows, int D, int scatter,
                                hipStream_t stream) {
  FOCR_CHECK_ARG(src && idx && dst && rows > 0 && D > 0, "bad argument");
  long g = (rows * D + 255) / 256;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)g), 256, 0, stream, src, idx, dst, rows, D, scatter);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------- cross-entropy
// loss = mean_r ( logsumexp(x[r]) - x[r][t_r] ), grad[r][c] = (softmax(x[r])[c] - [c == t_r]) / rows.
// C <= 64 classes: one lane per class, one wave per row; the batch mean is a fixed-order fold (deterministic).
__global__ __launch_bounds__(64) void ce_rows_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                     float* __restrict__ nll, float* __restrict__ grad, long rows,
                                                     int C) {
  const long r = blockIdx.x;
  const int c = threadIdx.x;
  const float v = c < C ? x[r * C + c] : -1e30f;
  const float mx = wave_max(v);
  const float e = c < C ? expf(v - mx) : 0.f;
  const float sum = wave_sum(e);
  const int t = (int)target[r];
  if (c < C) grad[r * C + c] = (e / sum - (c == t ? 1.f : 0.f)) / (float)rows;
  const float picked = wave_sum(c == t ? v : 0.f);
  if (c == 0) nll[r] = mx + logf(sum) - picked;
}
__global__ __launch_bounds__(64) void ce_fold_kernel(const float* __restrict__ nll, float* __restrict__ loss, long rows) {
  float acc = 0.f;
  for (long r = threadIdx.x; r < rows; r += 64) acc += nll[r];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = acc / (float)rows;
}
extern "C" int focr_cross_entropy_fwd(const float* logits, const long long* target, float* loss, float* nll_ws,
                                      float* grad, long rows, int C, hipStream_t stream) {
  FOCR_CHECK_ARG(logits && target && loss && nll_ws && grad && rows > 0 && C > 1 && C <= 64, "need 2 <= C <= 64");
  hipLaunchKernelGGL(ce_rows_kernel, dim3((int)rows), 64, 0, stream, logits, target, nll_ws, grad, rows, C);
  hipLaunchKernelGGL(ce_fold_kernel, dim3(1), 64, 0, stream, (const float*)nll_ws, loss, rows);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------- Adadelta
// torch.optim.Adadelta(lr, rho, eps): sq = rho sq + (1-rho) g^2; delta = sqrt(acc + eps) / sqrt(sq + eps) * g;
// acc = rho acc + (1-rho) delta^2; p -= lr * delta.  gscale folds the data-parallel 1/world averaging into g.
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ acc, long n,
                                                       float lr, float rho, float eps, float gscale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * gscale;
    const float s = rho * sq[i] + (1.f - rho) * gi * gi;
    const float d = sqrtf(acc[i] + eps) / sqrtf(s + eps) * gi;
    sq[i] = s;
    acc[i] = rho * acc[i] + (1.f - rho) * d * d;
    p[i] -= lr * d;
  }
}
extern "C" int focr_adadelta(float* p, const float* g, float* sq, float* acc, long n, float lr, float rho, float eps,
                             float gscale, hipStream_t stream) {
  FOCR_CHECK_ARG(p && g && sq && acc && n > 0, "bad argument");
  long gr = (n + 255) / 256;
  if (gr > 8192) gr = 8192;
  hipLaunchKernelGGL(adadelta_kernel, dim3((int)gr), 256, 0, stream, p, g, sq, acc, n, lr, rho, eps, gscale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------- attention
// q [B, Lq, H*Dk] (row pitch ldq), k / v [B, Lk, H*Dk] (pitch ldk); head h = columns h*Dk .. h*Dk + Dk - 1.
// The problems are tiny (B*H = 128 .. 512 heads, Lq <= ~40 queries, Lk <= 256 keys): what matters is to spread them
// over the chip.  One workgroup (256 threads) per (b, h, QUERY ROW): 4 .. 20 thousand blocks instead of B*H blocks that
// walk their rows one after the other (round 2, first version: 546 us forward / 1243 us backward at B = 32, 9 % and 12 % of
// the SLD step for 0.1 % of its flops).
// P (softmax, before dropout) and Pd (after dropout: what multiplies V and what the reference returns as the attention
// map) are written [B, H, Lq, Lk] for the backward.
// causal: key j visible to query i iff j <= i (subsequent_mask, transformer.py:203-207).
#define SA_LKMAX 256
template <int DK>
__global__ __launch_bounds__(256) void small_attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, float* __restrict__ O,
                                                             float* __restrict__ P, float* __restrict__ Pd, int H,
                                                             int Lq, int Lk, int ldq, int ldk, int ldo, float scale,
                                                             int causal, uint32_t drop_thr, float keep_scale,
                                                             uint64_t seed, const uint64_t* __restrict__ epoch) {
  seed = focr_epoch_seed(seed, epoch);
  __shared__ float qs[DK];
  __shared__ float ps[SA_LKMAX];
  __shared__ float red[4];
  const int b = blockIdx.x / H, h = blockIdx.x % H, i = blockIdx.y, tid = threadIdx.x;
  const float* qb = Q + (size_t)b * Lq * ldq + h * DK;
  const float* kb = K + (size_t)b * Lk * ldk + h * DK;
  const float* vb = V + (size_t)b * Lk * ldk + h * DK;
  float* ob = O + (size_t)b * Lq * ldo + h * DK;
  const size_t pbase = ((size_t)b * H + h) * Lq * Lk;
  for (int d = tid; d < DK; d += 256) qs[d] = qb[(size_t)i * ldq + d];
  __syncthreads();
  // scores: thread = key
  float s = -1e30f;
  if (tid < Lk && (!causal || tid <= i)) {
    const float4* kr = reinterpret_cast<const float4*>(kb + (size_t)tid * ldk);
    float acc = 0.f;
#pragma unroll 8
    for (int d4 = 0; d4 < DK / 4; ++d4) {
      const float4 kv = kr[d4];
      acc += qs[4 * d4] * kv.x + qs[4 * d4 + 1] * kv.y + qs[4 * d4 + 2] * kv.z + qs[4 * d4 + 3] * kv.w;
    }
    s = acc * scale;
  }
  float mx = wave_max(s);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float e = s > -1e29f ? expf(s - mx) : 0.f;
  const float sum = block_sum256(e, red);
  const float p = e / sum;
  float pd = p;
  if (drop_thr) {
    const uint32_t r = rng_hash(seed, (pbase + (size_t)i * Lk + tid)) >> 16;
    pd = r < drop_thr ? 0.f : p * keep_scale;
  }
  if (tid < Lk) {
    P[pbase + (size_t)i * Lk + tid] = p;
    Pd[pbase + (size_t)i * Lk + tid] = pd;
    ps[tid] = pd;
  }
  __syncthreads();
  // output: thread = head dim
  for (int d = tid; d < DK; d += 256) {
    float acc = 0.f;
    const int kend = causal ? i + 1 : Lk;
    for (int j = 0; j < kend; ++j) acc += ps[j] * vb[(size_t)j * ldk + d];
    ob[(size_t)i * ldo + d] = acc;
  }
}

// backward, pass 1: block (b, h, query row i) -> dS[i][:] = P (dropmask/keep * dPd - sum_j Pd dPd) with dPd = dO . V
// (+ the gradient arriving through the returned attention map), written to the workspace for pass 2, and dQ[i][:].
template <int DK>
__global__ __launch_bounds__(256) void small_attn_bwd_q_kernel(
    const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO, const float* __restrict__ P,
    const float* __restrict__ Pd, const float* __restrict__ dMap /* nullable: d loss / d Pd */, float* __restrict__ dQ,
    float* __restrict__ dS, int H, int Lq, int Lk, int ldq, int ldk, int ldo, float scale, int causal) {
  __shared__ float gs[DK];
  __shared__ float dss[SA_LKMAX];
  __shared__ float red[4];
  const int b = blockIdx.x / H, h = blockIdx.x % H, i = blockIdx.y, tid = threadIdx.x;
  const float* kb = K + (size_t)b * Lk * ldk + h * DK;
  const float* vb = V + (size_t)b * Lk * ldk + h * DK;
  const float* gb = dO + (size_t)b * Lq * ldo + h * DK;
  const size_t pbase = ((size_t)b * H + h) * Lq * Lk;
  for (int d = tid; d < DK; d += 256) gs[d] = gb[(size_t)i * ldo + d];
  __syncthreads();
  float dpd = 0.f, p = 0.f, pd = 0.f;
  const bool vis = tid < Lk && (!causal || tid <= i);
  if (vis) {
    const float4* vr = reinterpret_cast<const float4*>(vb + (size_t)tid * ldk);
    float acc = 0.f;
#pragma unroll 8
    for (int d4 = 0; d4 < DK / 4; ++d4) {
      const float4 vv = vr[d4];
      acc += gs[4 * d4] * vv.x + gs[4 * d4 + 1] * vv.y + gs[4 * d4 + 2] * vv.z + gs[4 * d4 + 3] * vv.w;
    }
    dpd = acc;
    if (dMap) dpd += dMap[pbase + (size_t)i * Lk + tid];      // the attention map is an output too (text-focus L1 term)
    p = P[pbase + (size_t)i * Lk + tid];
    pd = Pd[pbase + (size_t)i * Lk + tid];
  }
  const float D = block_sum256(pd * dpd, red);
  // dP = dPd * (Pd / P)  (Pd = P * mask / keep; P > 0 for every visible key)
  const float dp = (vis && p > 0.f) ? dpd * (pd / p) : 0.f;
  const float dsv = vis ? p * (dp - D) : 0.f;
  if (tid < Lk) {
    dS[pbase + (size_t)i * Lk + tid] = dsv;
    dss[tid] = dsv;
  }
  __syncthreads();
  for (int d = tid; d < DK; d += 256) {                      // dQ row: thread = head dim
    float acc = 0.f;
    const int kend = causal ? i + 1 : Lk;
    for (int j = 0; j < kend; ++j) acc += dss[j] * kb[(size_t)j * ldk + d];
    dQ[(size_t)b * Lq * ldq + h * DK + (size_t)i * ldq + d] = acc * scale;
  }
}
// ---- DK = 64 (the 16 x 64 cross attention of the text- / stroke-focus recognizers over 256 image positions), round 6:
// one block per (batch, head) walks over ALL query rows.  The per-row kernels above re-read the head's K and V (2 x 64 KB)
// from L2 for every one of the ~11 query rows of every (batch, head): 2.9 GB per launch at B = 128, 255 us forward /
// 479 us backward-q (profiles/r06b_tfl_bygrid.txt).  Here a thread keeps its key's K (forward) / V (backward) row in
// registers for the whole launch and the other matrix lives in LDS (64 KB).  Scores, softmax, dropout bits and the
// per-element formulas are those of the per-row kernels; the output sums run over four key quarters in a fixed order.
#define SA64_LDS ((SA_LKMAX * 64 + 64 + SA_LKMAX + 4 * 64 + 8) * 4)
__global__ __launch_bounds__(256) void small_attn_fwd_rows64_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, float* __restrict__ O,
    float* __restrict__ P, float* __restrict__ Pd, int H, int Lq, int Lk, int ldq, int ldk, int ldo, float scale, int causal,
    uint32_t drop_thr, float keep_scale, uint64_t seed, const uint64_t* __restrict__ epoch) {
  seed = focr_epoch_seed(seed, epoch);
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  float* Vs = sa_sm;                         // [Lk][64]
  float* qs = Vs + SA_LKMAX * 64;            // [64]
  float* ps = qs + 64;                       // [SA_LKMAX]
  float* parts = ps + SA_LKMAX;              // [4][64]
  float* red = parts + 4 * 64;               // [4 (+4)]
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const float* qb = Q + (size_t)b * Lq * ldq + h * 64;
  const float* kb = K + (size_t)b * Lk * ldk + h * 64;
  const float* vb = V + (size_t)b * Lk * ldk + h * 64;
  float* ob = O + (size_t)b * Lq * ldo + h * 64;
  const size_t pbase = ((size_t)b * H + h) * Lq * Lk;
  float4 kr[16];
#pragma unroll
  for (int d4 = 0; d4 < 16; ++d4)
    kr[d4] = tid < Lk ? reinterpret_cast<const float4*>(kb + (size_t)tid * ldk)[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int idx = tid; idx < Lk * 16; idx += 256)
    reinterpret_cast<float4*>(Vs)[idx] = reinterpret_cast<const float4*>(vb + (size_t)(idx >> 4) * ldk)[idx & 15];
  const int d = tid & 63, part = tid >> 6;
  for (int i = 0; i < Lq; ++i) {
    __syncthreads();                                       // Vs ready (first row) / the previous row's parts consumed
    if (tid < 64) qs[tid] = qb[(size_t)i * ldq + tid];
    __syncthreads();
    float s = -1e30f;
    if (tid < Lk && (!causal || tid <= i)) {
      float acc = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < 16; ++d4)
        acc += qs[4 * d4] * kr[d4].x + qs[4 * d4 + 1] * kr[d4].y + qs[4 * d4 + 2] * kr[d4].z + qs[4 * d4 + 3] * kr[d4].w;
      s = acc * scale;
    }
    float mx = wave_max(s);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();                                       // red is reused by block_sum256
    const float e = s > -1e29f ? expf(s - mx) : 0.f;
    const float sum = block_sum256(e, red);
    const float p = e / sum;
    float pd = p;
    if (drop_thr) {
      const uint32_t r = rng_hash(seed, (pbase + (size_t)i * Lk + tid)) >> 16;
      pd = r < drop_thr ? 0.f : p * keep_scale;
    }
    if (tid < Lk) {
      P[pbase + (size_t)i * Lk + tid] = p;
      Pd[pbase + (size_t)i * Lk + tid] = pd;
      ps[tid] = pd;
    }
    __syncthreads();
    const int kend = causal ? i + 1 : Lk;
    float acc = 0.f;
    for (int j = part * 64; j < min(kend, part * 64 + 64); ++j) acc += ps[j] * Vs[j * 64 + d];
    parts[part * 64 + d] = acc;
    __syncthreads();
    if (tid < 64) ob[(size_t)i * ldo + tid] = (parts[tid] + parts[64 + tid]) + (parts[128 + tid] + parts[192 + tid]);
  }
}

__global__ __launch_bounds__(256) void small_attn_bwd_q_rows64_kernel(
    const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO, const float* __restrict__ P,
    const float* __restrict__ Pd, const float* __restrict__ dMap, float* __restrict__ dQ, float* __restrict__ dS, int H,
    int Lq, int Lk, int ldq, int ldk, int ldo, float scale, int causal) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  float* Ks = sa_sm;                         // [Lk][64]
  float* gs = Ks + SA_LKMAX * 64;            // [64]
  float* dss = gs + 64;                      // [SA_LKMAX]
  float* parts = dss + SA_LKMAX;             // [4][64]
  float* red = parts + 4 * 64;
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const float* kb = K + (size_t)b * Lk * ldk + h * 64;
  const float* vb = V + (size_t)b * Lk * ldk + h * 64;
  const float* gb = dO + (size_t)b * Lq * ldo + h * 64;
  const size_t pbase = ((size_t)b * H + h) * Lq * Lk;
  float4 vr[16];
#pragma unroll
  for (int d4 = 0; d4 < 16; ++d4)
    vr[d4] = tid < Lk ? reinterpret_cast<const float4*>(vb + (size_t)tid * ldk)[d4] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int idx = tid; idx < Lk * 16; idx += 256)
    reinterpret_cast<float4*>(Ks)[idx] = reinterpret_cast<const float4*>(kb + (size_t)(idx >> 4) * ldk)[idx & 15];
  const int d = tid & 63, part = tid >> 6;
  for (int i = 0; i < Lq; ++i) {
    __syncthreads();
    if (tid < 64) gs[tid] = gb[(size_t)i * ldo + tid];
    __syncthreads();
    float dpd = 0.f, p = 0.f, pd = 0.f;
    const bool vis = tid < Lk && (!causal || tid <= i);
    if (vis) {
      float acc = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < 16; ++d4)
        acc += gs[4 * d4] * vr[d4].x + gs[4 * d4 + 1] * vr[d4].y + gs[4 * d4 + 2] * vr[d4].z + gs[4 * d4 + 3] * vr[d4].w;
      dpd = acc;
      if (dMap) dpd += dMap[pbase + (size_t)i * Lk + tid];
      p = P[pbase + (size_t)i * Lk + tid];
      pd = Pd[pbase + (size_t)i * Lk + tid];
    }
    const float D = block_sum256(pd * dpd, red);
    const float dp = (vis && p > 0.f) ? dpd * (pd / p) : 0.f;
    const float dsv = vis ? p * (dp - D) : 0.f;
    if (tid < Lk) {
      dS[pbase + (size_t)i * Lk + tid] = dsv;
      dss[tid] = dsv;
    }
    __syncthreads();
    const int kend = causal ? i + 1 : Lk;
    float acc = 0.f;
    for (int j = part * 64; j < min(kend, part * 64 + 64); ++j) acc += dss[j] * Ks[j * 64 + d];
    parts[part * 64 + d] = acc;
    __syncthreads();
    if (tid < 64)
      dQ[(size_t)b * Lq * ldq + h * 64 + (size_t)i * ldq + tid] =
          ((parts[tid] + parts[64 + tid]) + (parts[128 + tid] + parts[192 + tid])) * scale;
  }
}
static bool sa_rows64_ready() {
  static focr_dev_flags attr;
  if (focr_dev_first(attr)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(small_attn_fwd_rows64_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SA64_LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(small_attn_bwd_q_rows64_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, SA64_LDS) != hipSuccess)
      return false;
    focr_dev_mark(attr);
  }
  return true;
}

// backward, pass 2: block (b, h, SA_JT keys) -> dK[j][:], dV[j][:] (sums over the query rows in index order)
#define SA_JT 8
template <int DK>
__global__ __launch_bounds__(256) void small_attn_bwd_kv_kernel(
    const float* __restrict__ Q, const float* __restrict__ dO, const float* __restrict__ Pd, const float* __restrict__ dS,
    float* __restrict__ dK, float* __restrict__ dV, int H, int Lq, int Lk, int ldq, int ldk, int ldo, float scale,
    int causal) {
  const int b = blockIdx.x / H, h = blockIdx.x % H, j0 = blockIdx.y * SA_JT, tid = threadIdx.x;
  const float* qb = Q + (size_t)b * Lq * ldq + h * DK;
  const float* gb = dO + (size_t)b * Lq * ldo + h * DK;
  const size_t pbase = ((size_t)b * H + h) * Lq * Lk;
  for (int idx = tid; idx < SA_JT * DK; idx += 256) {
    const int j = j0 + idx / DK, d = idx % DK;
    if (j >= Lk) continue;
    float ak = 0.f, av = 0.f;
    for (int i = causal ? j : 0; i < Lq; ++i) {
      ak += dS[pbase + (size_t)i * Lk + j] * qb[(size_t)i * ldq + d];
      av += Pd[pbase + (size_t)i * Lk + j] * gb[(size_t)i * ldo + d];
    }
    dK[(size_t)b * Lk * ldk + h * DK + (size_t)j * ldk + d] = ak * scale;
    dV[(size_t)b * Lk * ldk + h * DK + (size_t)j * ldk + d] = av;
  }
}

extern "C" int focr_small_attention_fwd(const float* q, const float* k, const float* v, float* o, float* p, float* pd,
                                        int B, int H, int Lq, int Lk, int Dk, int ldq, int ldk, int ldo, float scale,
                                        int causal, float p_drop, uint64_t seed, hipStream_t stream) {
  FOCR_CHECK_ARG(q && k && v && o && p && pd, "null pointer");
  FOCR_CHECK_ARG((Dk == 256 || Dk == 64) && Lk > 0 && Lk <= SA_LKMAX && Lq > 0 && ldk % 4 == 0,
                 "need Dk in {64, 256}, Lk <= 256");
  FOCR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && Lq <= 65535, "bad dropout probability / too many query rows");
  const uint32_t thr = (uint32_t)(p_drop * 65536.f + 0.5f);
  FOCR_CHECK_ARG(thr < 65536u, "dropout probability rounds to 1");
  const float ks = thr ? 65536.f / (65536.f - (float)thr) : 1.f;
  if (Dk == 64 && ldq % 4 == 0 && sa_rows64_ready())
    hipLaunchKernelGGL(small_attn_fwd_rows64_kernel, dim3(B * H), 256, SA64_LDS, stream, q, k, v, o, p, pd, H, Lq, Lk, ldq,
                       ldk, ldo, scale, causal, thr, ks, seed, focr_seed_epoch());
  else if (Dk == 64)
    hipLaunchKernelGGL((small_attn_fwd_kernel<64>), dim3(B * H, Lq), 256, 0, stream, q, k, v, o, p, pd, H, Lq, Lk, ldq,
                       ldk, ldo, scale, causal, thr, ks, seed, focr_seed_epoch());
  else
    hipLaunchKernelGGL((small_attn_fwd_kernel<256>), dim3(B * H, Lq), 256, 0, stream, q, k, v, o, p, pd, H, Lq, Lk, ldq,
                       ldk, ldo, scale, causal, thr, ks, seed, focr_seed_epoch());
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// ws: B*H*Lq*Lk floats
extern "C" int focr_small_attention_bwd(const float* q, const float* k, const float* v, const float* d_o,
                                        const float* p, const float* pd, const float* dmap, float* dq, float* dk,
                                        float* dv, float* ws, int B, int H, int Lq, int Lk, int Dk, int ldq, int ldk,
                                        int ldo, float scale, int causal, hipStream_t stream) {
  FOCR_CHECK_ARG(q && k && v && d_o && p && pd && dq && dk && dv && ws, "null pointer");
  FOCR_CHECK_ARG((Dk == 256 || Dk == 64) && Lk > 0 && Lk <= SA_LKMAX && Lq > 0 && ldk % 4 == 0,
                 "need Dk in {64, 256}, Lk <= 256");
  FOCR_CHECK_ARG(Lq <= 65535, "too many query rows for the launch grid");
  const dim3 gq(B * H, Lq), gkv(B * H, (Lk + SA_JT - 1) / SA_JT);
  if (Dk == 64) {
    if (sa_rows64_ready())
      hipLaunchKernelGGL(small_attn_bwd_q_rows64_kernel, dim3(B * H), 256, SA64_LDS, stream, k, v, d_o, p, pd, dmap, dq, ws, H,
                         Lq, Lk, ldq, ldk, ldo, scale, causal);
    else
      hipLaunchKernelGGL((small_attn_bwd_q_kernel<64>), gq, 256, 0, stream, k, v, d_o, p, pd, dmap, dq, ws, H, Lq, Lk, ldq,
                         ldk, ldo, scale, causal);
    hipLaunchKernelGGL((small_attn_bwd_kv_kernel<64>), gkv, 256, 0, stream, q, d_o, pd, (const float*)ws, dk, dv, H, Lq, Lk,
                       ldq, ldk, ldo, scale, causal);
  } else {
    hipLaunchKernelGGL((small_attn_bwd_q_kernel<256>), gq, 256, 0, stream, k, v, d_o, p, pd, dmap, dq, ws, H, Lq, Lk, ldq,
                       ldk, ldo, scale, causal);
    hipLaunchKernelGGL((small_attn_bwd_kv_kernel<256>), gkv, 256, 0, stream, q, d_o, pd, (const float*)ws, dk, dv, H, Lq, Lk,
                       ldq, ldk, ldo, scale, causal);
  }
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------- text-focus loss
// nn.L1Loss (mean |a - b|) between two attention maps (loss/text_focus_loss.py:92) and its gradient w.r.t. b
// (the HR map is a constant); fixed-order reduction.
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ part, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += fabsf(a[i] - b[i]);
  s = block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void l1_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int nb,
                                                     float inv_n) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s * inv_n;
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ g, float* __restrict__ db, long n) {
  const float k = g[0] / (float)n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = b[i] - a[i];
    db[i] = d > 0.f ? k : (d < 0.f ? -k : 0.f);
  }
}
#define L1_BLOCKS 256
extern "C" int focr_l1_fwd(const float* a, const float* b, float* out, float* ws, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && out && ws && n > 0, "bad argument (ws: 256 floats)");
  hipLaunchKernelGGL(l1_partial_kernel, dim3(L1_BLOCKS), 256, 0, stream, a, b, ws, n);
  hipLaunchKernelGGL(l1_fold_kernel, dim3(1), 64, 0, stream, (const float*)ws, out, L1_BLOCKS, 1.f / (float)n);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_l1_bwd(const float* a, const float* b, const float* g, float* db, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && g && db && n > 0, "bad argument");
  long gr = (n + 255) / 256;
  if (gr > 2048) gr = 2048;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3((int)gr), 256, 0, stream, a, b, g, db, n);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// weight_cross_entropy (loss/weight_ce_loss.py:38-45): -mean_r log( w[t_r][t_r] e^{x[r][t_r]} / sum_c w[t_r][c] e^{x[r][c]} )
// with a [C][C] confusion-derived weight table; grad[r][c] = (w[t][c] e^{x_c} / S - [c == t]) / rows.
__global__ __launch_bounds__(64) void wce_rows_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                      const float* __restrict__ table, float* __restrict__ nll,
                                                      float* __restrict__ grad, long rows, int C) {
  const long r = blockIdx.x;
  const int c = threadIdx.x, t = (int)target[r];
  const float v = c < C ? x[r * C + c] : -1e30f;
  const float mx = wave_max(v);                                     // stabilised: the ratio is shift-invariant
  const float e = c < C ? table[t * C + c] * expf(v - mx) : 0.f;
  const float sum = wave_sum(e);
  if (c < C) grad[r * C + c] = (e / sum - (c == t ? 1.f : 0.f)) / (float)rows;
  const float picked = wave_sum(c == t ? e : 0.f);
  if (c == 0) nll[r] = -logf(picked / sum);
}
extern "C" int focr_weight_cross_entropy_fwd(const float* logits, const long long* target, const float* table,
                                             float* loss, float* nll_ws, float* grad, long rows, int C,
                                             hipStream_t stream) {
  FOCR_CHECK_ARG(logits && target && table && loss && nll_ws && grad && rows > 0 && C > 1 && C <= 64, "need 2 <= C <= 64");
  hipLaunchKernelGGL(wce_rows_kernel, dim3((int)rows), 64, 0, stream, logits, target, table, nll_ws, grad, rows, C);
  hipLaunchKernelGGL(ce_fold_kernel, dim3(1), 64, 0, stream, (const float*)nll_ws, loss, rows);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ---- the same two losses on PADDED label layouts (round 6): what depends on the batch's labels is data, not shape.
// The focus losses' recognizer decodes [B, L] teacher-forcing positions with L = the longest label of the batch
// (text_focus_loss.py:62-81), its attention-map L1 is a mean over all B x 16 x L x 256 entries and the cross entropy a mean over
// the sum(len) real positions -- so every launch of the step changes shape with the labels and the step could not be recorded
// (csrc/replay.hip).  Here L is a capacity (labels padded to a bucket), and a three-word device plan carries the real
// numbers: plan[0] = longest label, plan[1] = number of real positions.  Padded decoder positions see the causal mask, so
// they do not touch the real ones; these kernels leave them out of the sums, the means and the gradients.
// a, b: [outer][L][inner]; positions j >= plan[0] do not count; mean over outer * plan[0] * inner entries
__global__ __launch_bounds__(256) void l1_masked_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                float* __restrict__ part, long n, int L, int inner,
                                                                const long long* __restrict__ plan) {
  __shared__ float red[4];
  const int lmax = (int)plan[0];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    if ((int)((i / inner) % L) < lmax) s += fabsf(a[i] - b[i]);
  s = block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void l1_masked_fold_kernel(const float* __restrict__ part, float* __restrict__ out, int nb,
                                                            long outer_inner, const long long* __restrict__ plan) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = s / ((float)outer_inner * (float)plan[0]);
}
__global__ __launch_bounds__(256) void l1_masked_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ g, float* __restrict__ db, long n,
                                                            int L, int inner, long outer_inner,
                                                            const long long* __restrict__ plan) {
  const int lmax = (int)plan[0];
  const float k = g[0] / ((float)outer_inner * (float)lmax);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = b[i] - a[i];
    db[i] = (int)((i / inner) % L) < lmax ? (d > 0.f ? k : (d < 0.f ? -k : 0.f)) : 0.f;
  }
}
extern "C" int focr_l1_masked_fwd(const float* a, const float* b, float* out, float* ws, long outer, int L, int inner,
                                  const long long* plan, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && out && ws && plan && outer > 0 && L > 0 && inner > 0, "bad argument (ws: 256 floats)");
  const long n = outer * L * inner;
  hipLaunchKernelGGL(l1_masked_partial_kernel, dim3(L1_BLOCKS), 256, 0, stream, a, b, ws, n, L, inner, plan);
  hipLaunchKernelGGL(l1_masked_fold_kernel, dim3(1), 64, 0, stream, (const float*)ws, out, L1_BLOCKS, outer * inner, plan);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_l1_masked_bwd(const float* a, const float* b, const float* g, float* db, long outer, int L, int inner,
                                  const long long* plan, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && g && db && plan && outer > 0 && L > 0 && inner > 0, "bad argument");
  const long n = outer * L * inner;
  long gr = (n + 255) / 256;
  if (gr > 2048) gr = 2048;
  hipLaunchKernelGGL(l1_masked_bwd_kernel, dim3((int)gr), 256, 0, stream, a, b, g, db, n, L, inner, outer * inner, plan);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// weight_cross_entropy over padded rows: target < 0 marks a padded position (no loss, zero gradient); mean over plan[1] rows
__global__ __launch_bounds__(64) void wce_rows_masked_kernel(const float* __restrict__ x, const long long* __restrict__ target,
                                                             const float* __restrict__ table, float* __restrict__ nll,
                                                             float* __restrict__ grad, int C,
                                                             const long long* __restrict__ plan) {
  const long r = blockIdx.x;
  const int c = threadIdx.x, t = (int)target[r];
  if (t < 0) {                                                        // (block-uniform)
    if (c < C) grad[r * C + c] = 0.f;
    if (c == 0) nll[r] = 0.f;
    return;
  }
  const float inv = 1.f / (float)plan[1];
  const float v = c < C ? x[r * C + c] : -1e30f;
  const float mx = wave_max(v);
  const float e = c < C ? table[t * C + c] * expf(v - mx) : 0.f;
  const float sum = wave_sum(e);
  if (c < C) grad[r * C + c] = (e / sum - (c == t ? 1.f : 0.f)) * inv;
  const float picked = wave_sum(c == t ? e : 0.f);
  if (c == 0) nll[r] = -logf(picked / sum);
}
__global__ __launch_bounds__(64) void ce_masked_fold_kernel(const float* __restrict__ nll, float* __restrict__ loss, long rows,
                                                            const long long* __restrict__ plan) {
  float acc = 0.f;
  for (long r = threadIdx.x; r < rows; r += 64) acc += nll[r];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = acc / (float)plan[1];
}
extern "C" int focr_weight_cross_entropy_masked_fwd(const float* logits, const long long* target, const float* table,
                                                    float* loss, float* nll_ws, float* grad, long rows, int C,
                                                    const long long* plan, hipStream_t stream) {
  FOCR_CHECK_ARG(logits && target && table && loss && nll_ws && grad && plan && rows > 0 && C > 1 && C <= 64,
                 "need 2 <= C <= 64");
  hipLaunchKernelGGL(wce_rows_masked_kernel, dim3((int)rows), 64, 0, stream, logits, target, table, nll_ws, grad, C, plan);
  hipLaunchKernelGGL(ce_masked_fold_kernel, dim3(1), 64, 0, stream, (const float*)nll_ws, loss, rows, plan);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

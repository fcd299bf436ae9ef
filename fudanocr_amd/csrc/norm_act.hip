// Memory-bound kernels of the SR trunk: BatchNorm (train / eval, fwd / bwd, fused activation
// and residual), the reference's own LayerNorm, PReLU, pixel-shuffle + mish, tanh + layout,
// positional-encoding concat, dropout, MSE.  All tensors fp32, channel-last (NHWC); every
// kernel is HBM-bound and uses 16-byte accesses where the layout allows.
//
// Reference semantics (scene-text-telescope/):
//   BatchNorm2d/1d  torch semantics, SURVEY.md Appendix C   (model/tsrn.py:81-86, stn_head.py:17-21)
//   LayerNorm       unbiased std, eps added to std          (model/tbsrn.py:23-36)
//   mish            x*tanh(softplus(x)), threshold 20       (model/tsrn.py:117-125)
//   PReLU           single learnable slope                  (model/tsrn.py:28)
//   PixelShuffle(2) out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w]  (model/tsrn.py:101-114)
#include "focr_common.h"

#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_MISH 4

__device__ __forceinline__ float act_fwd(float x, int act) {
  if (act == ACT_RELU) return fmaxf(x, 0.f);
  if (act == ACT_MISH) return mish_f(x);
  return x;
}
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == ACT_MISH) return mish_grad_f(x);
  return 1.f;
}

// ---------------------------------------------------------------------------------------
// column reductions over [rows][C] (C % 4 == 0 not required): block = 64 columns x 4 row lanes
// mode 0: sum x          mode 1: sum (x - mean)^2, mean = sum0[c]/rows
// ---------------------------------------------------------------------------------------
// Vectorised: thread = 4 consecutive channels (float4), C4 = C/4 threads per row, 256/C4 rows per pass.
__global__ __launch_bounds__(256) void bn_colstat_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ sum0,
                                                         float* __restrict__ out, long rows, int C,
                                                         int mode) {
  __shared__ float4 red[256];
  const int C4 = C >> 2;
  // channel groups handled by this block: blockIdx.x covers min(C4, 256) float4 columns
  const int cols = C4 < 256 ? C4 : 256;            // float4 columns per block
  const int rl_n = 256 / cols;                     // row lanes
  const int col = blockIdx.x * cols + (threadIdx.x % cols);
  const int rl = threadIdx.x / cols;
  long rows_per = (rows + gridDim.y - 1) / gridDim.y;
  long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < C4 && rl < rl_n) {
    float4 mean = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode) {   // sum0 = folded column sums of pass 0
      float4 t = *reinterpret_cast<const float4*>(sum0 + col * 4);
      float ir = 1.f / (float)rows;
      mean = make_float4(t.x * ir, t.y * ir, t.z * ir, t.w * ir);
    }
    for (long r = r0 + rl; r < r1; r += rl_n) {
      float4 v = *reinterpret_cast<const float4*>(x + r * C + col * 4);
      v.x -= mean.x; v.y -= mean.y; v.z -= mean.z; v.w -= mean.w;
      if (mode) { s.x += v.x * v.x; s.y += v.y * v.y; s.z += v.z * v.z; s.w += v.w * v.w; }
      else { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && col < C4) {
    for (int j = 1; j < rl_n; ++j) {
      float4 t = red[j * cols + (threadIdx.x % cols)];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    // deterministic: slab partials, summed in fixed order by bn_fold_kernel (no atomics in the forward)
    *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * C + col * 4) = s;
  }
}

// out[c] = sum over slabs of part[slab][c]: one block per 4 channels, 256 threads stride the slabs, then a
// fixed-order LDS tree -> deterministic and parallel
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                      int slabs, int C) {
  __shared__ float4 red[256];
  const int c4 = blockIdx.x;                       // float4 column
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = threadIdx.x; j < slabs; j += 256) {
    float4 v = *reinterpret_cast<const float4*>(part + (size_t)j * C + c4 * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      float4 a = red[threadIdx.x], b = red[threadIdx.x + w];
      red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *reinterpret_cast<float4*>(out + c4 * 4) = red[0];
}

// fold of two partial arrays in one launch (backward: dbeta and dgamma): blockIdx.y selects the array
__global__ __launch_bounds__(256) void bn_fold2_kernel(const float* __restrict__ pa, float* __restrict__ oa,
                                                       const float* __restrict__ pb, float* __restrict__ ob,
                                                       int slabs, int C) {
  __shared__ float4 red[256];
  const float* part = blockIdx.y ? pb : pa;
  float* out = blockIdx.y ? ob : oa;
  const int c4 = blockIdx.x;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j0 = threadIdx.x; j0 < slabs; j0 += 8 * 256) {       // eight loads in flight per thread (was one: the fold of
    float4 v[8];                                                 // 2048 slabs took as long as the reduction before it)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = j0 + 256 * u;
      v[u] = j < slabs ? *reinterpret_cast<const float4*>(part + (size_t)j * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      float4 a = red[threadIdx.x], b = red[threadIdx.x + w];
      red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *reinterpret_cast<float4*>(out + c4 * 4) = red[0];
}

// fold of the centred sum-of-squares partials (same tree as bn_fold_kernel) + the finalize step, one launch
__global__ __launch_bounds__(256) void bn_fold_finalize_kernel(
    const float* __restrict__ part, const float* __restrict__ sum, float* __restrict__ sq,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ rmean,
    float* __restrict__ rvar, long long* nbt, long rows, int slabs, int C, float momentum, float eps) {
  __shared__ float4 red[256];
  const int c4 = blockIdx.x;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = threadIdx.x; j < slabs; j += 256) {
    float4 v = *reinterpret_cast<const float4*>(part + (size_t)j * C + c4 * 4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      float4 a = red[threadIdx.x], b = red[threadIdx.x + w];
      red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) {
    const int c = c4 * 4 + threadIdx.x;
    const float4 r = red[0];
    const float q = threadIdx.x == 0 ? r.x : threadIdx.x == 1 ? r.y : threadIdx.x == 2 ? r.z : r.w;
    sq[c] = q;
    float mean = sum[c] / (float)rows;
    float var = q / (float)rows;
    save_mean[c] = mean;
    save_invstd[c] = rsqrtf(var + eps);
    if (rmean) {
      float unb = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
    if (c == 0 && nbt) *nbt += 1;
  }
}


// y = act(gamma * (x - mean) * invstd + beta) + residual       (total = rows*C elements)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ res,
                                                       float* __restrict__ y, long total4, int C,
                                                       int act) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long)gridDim.x * blockDim.x) {
    int c = (int)((i * 4) % C);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 g = *reinterpret_cast<const float4*>(gamma + c);
    float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 mu = *reinterpret_cast<const float4*>(mean + c);
    float4 is = *reinterpret_cast<const float4*>(invstd + c);
    float4 o;
    o.x = act_fwd(g.x * (v.x - mu.x) * is.x + b.x, act);
    o.y = act_fwd(g.y * (v.y - mu.y) * is.y + b.y, act);
    o.z = act_fwd(g.z * (v.z - mu.z) * is.z + b.z, act);
    o.w = act_fwd(g.w * (v.w - mu.w) * is.w + b.w, act);
    if (res) {
      float4 r = reinterpret_cast<const float4*>(res)[i];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// backward pass 1: sum_g[c] = sum g, sum_gx[c] = sum g * xhat, g = dz * act'(gamma*xhat+beta); float4 columns
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ sum_g, float* __restrict__ sum_gx, long rows, int C, int act, int lddz) {
  __shared__ float4 red0[256], red1[256];
  const int C4 = C >> 2;
  const int cols = C4 < 256 ? C4 : 256;
  const int rl_n = 256 / cols;
  const int col = blockIdx.x * cols + (threadIdx.x % cols);
  const int rl = threadIdx.x / cols;
  long rows_per = (rows + gridDim.y - 1) / gridDim.y;
  long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < C4 && rl < rl_n) {
    float4 g4 = *reinterpret_cast<const float4*>(gamma + col * 4), b4 = *reinterpret_cast<const float4*>(beta + col * 4);
    float4 m4 = *reinterpret_cast<const float4*>(mean + col * 4), i4 = *reinterpret_cast<const float4*>(invstd + col * 4);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
    const float* xp = x + (r0 + rl) * C + col * 4;
    const float* dp = dz + (r0 + rl) * lddz + col * 4;
    const long xstep = (long)rl_n * C, dstep = (long)rl_n * lddz;
    // four rows per iteration: the eight loads are issued together (the kernel was latency-bound with one row pair in
    // flight per thread: 3.4 TB/s); rows past the slab end are clamped to the last row and weighted 0
    for (long r = r0 + rl; r < r1; r += 4 * rl_n, xp += 4 * xstep, dp += 4 * dstep) {
      float4 xv[4], dv[4];
      float wt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = r + (long)u * rl_n < r1;
        wt[u] = ok ? 1.f : 0.f;
        xv[u] = *reinterpret_cast<const float4*>(ok ? xp + u * xstep : xp);
        dv[u] = *reinterpret_cast<const float4*>(ok ? dp + u * dstep : dp);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh = (xx[e] - mm[e]) * ii[e];
          float g = wt[u] * dd[e] * act_grad(gg[e] * xh + bb[e], act);
          s0[e] += g;
          s1[e] += g * xh;
        }
      }
    }
  }
  red0[threadIdx.x] = make_float4(s0[0], s0[1], s0[2], s0[3]);
  red1[threadIdx.x] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  __syncthreads();
  if (rl == 0 && col < C4) {
    float4 a = red0[threadIdx.x], b = red1[threadIdx.x];
    for (int j = 1; j < rl_n; ++j) {
      float4 t = red0[j * cols + (threadIdx.x % cols)], u = red1[j * cols + (threadIdx.x % cols)];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      b.x += u.x; b.y += u.y; b.z += u.z; b.w += u.w;
    }
    // slab partials (folded in fixed order by bn_fold_kernel): no same-address atomic serialisation
    *reinterpret_cast<float4*>(sum_g + (size_t)blockIdx.y * C + col * 4) = a;
    *reinterpret_cast<float4*>(sum_gx + (size_t)blockIdx.y * C + col * 4) = b;
  }
}

// backward pass 2 (train): dx = gamma*invstd*(g - sum_g/rows - xhat*sum_gx/rows)
//          (eval, sums == nullptr): dx = gamma*invstd*g
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ sum_g, const float* __restrict__ sum_gx, float* __restrict__ dx, long total4,
    long rows, int C, int act, int lddz) {
  const float inv_rows = 1.f / (float)rows;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, step = (long)gridDim.x * blockDim.x;
  // the grid stride is a multiple of C (ew_grid() hands out multiples of 256 threads = 1024 floats): every thread
  // keeps its four channels for the whole loop, so the per-channel coefficients are loaded once
  //   xh = (x - mo)*is ; y = ga*xh + be ; g = dz*act'(y) ; dx = gi*(g - sg - xh*sgx)
  const bool fixed_c = (step * 4) % C == 0;
  int c = (int)((i0 * 4) % C);
  float is[4], mo[4], ga[4], be[4], gi[4], sg[4], sgx[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    is[e] = invstd[c + e];
    mo[e] = mean[c + e];
    ga[e] = gamma[c + e];
    be[e] = beta[c + e];
    gi[e] = ga[e] * is[e];
    sg[e] = sum_g ? sum_g[c + e] * inv_rows : 0.f;
    sgx[e] = sum_g ? sum_gx[c + e] * inv_rows : 0.f;
  }
  for (long i = i0; i < total4; i += step) {
    if (!fixed_c) {
      c = (int)((i * 4) % C);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        is[e] = invstd[c + e];
        mo[e] = mean[c + e];
        ga[e] = gamma[c + e];
        be[e] = beta[c + e];
        gi[e] = ga[e] * is[e];
        sg[e] = sum_g ? sum_g[c + e] * inv_rows : 0.f;
        sgx[e] = sum_g ? sum_gx[c + e] * inv_rows : 0.f;
      }
    }
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 d;
    if (lddz == C) {
      d = reinterpret_cast<const float4*>(dz)[i];
    } else {                                   // dz is a column slice of a wider row-major matrix (row pitch lddz)
      const long row = (i * 4) / C;
      d = *reinterpret_cast<const float4*>(dz + row * lddz + (i * 4 - row * C));
    }
    float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w}, oo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh = (vv[e] - mo[e]) * is[e];
      float g = dd[e] * act_grad(ga[e] * xh + be[e], act);
      oo[e] = gi[e] * (g - sg[e] - xh * sgx[e]);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(oo[0], oo[1], oo[2], oo[3]);
  }
}

__global__ void bn_eval_prep_kernel(const float* __restrict__ rvar, float* __restrict__ invstd, int C,
                                    float eps) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) invstd[c] = rsqrtf(rvar[c] + eps);
}

// ---------------------------------------------------------------------------------------
// LayerNorm of the reference (unbiased std, eps on the std), D = 128, one wave per row
// ---------------------------------------------------------------------------------------
// Half a wave (32 lanes x float4) owns one row of D = 128: every load/store instruction of a wave moves two
// full rows (1 KB) and the row statistics are 32-lane butterflies.
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int D>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ res,
                                                     const float* __restrict__ a,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     float* __restrict__ save_mean,
                                                     float* __restrict__ save_rinv, long rows, float eps) {
  static_assert(D == 128, "half-wave rows need D == 128");
  const int li = threadIdx.x & 31;
  long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= rows) return;
  float4 v = reinterpret_cast<const float4*>(x + row * D)[li];
  if (res) {
    float4 r = reinterpret_cast<const float4*>(res + row * D)[li];
    v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
  }
  const float4 av = reinterpret_cast<const float4*>(a)[li], bv = reinterpret_cast<const float4*>(b)[li];
  float mean = half_sum((v.x + v.y) + (v.z + v.w)) / D;
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  float sd = sqrtf(half_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) / (D - 1));
  float rinv = 1.f / (sd + eps);
  float4 o;
  o.x = av.x * v.x * rinv + bv.x;
  o.y = av.y * v.y * rinv + bv.y;
  o.z = av.z * v.z * rinv + bv.z;
  o.w = av.w * v.w * rinv + bv.w;
  reinterpret_cast<float4*>(y + row * D)[li] = o;
  if (li == 0) {
    save_mean[row] = mean;
    save_rinv[row] = rinv;
  }
}

// xin = x (+res) is recomputed from the same inputs.  dx is also the gradient of `res`.
template <int D>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ res,
                                                     const float* __restrict__ a,
                                                     const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_rinv,
                                                     float* __restrict__ dx, float* __restrict__ da,
                                                     float* __restrict__ db, long rows, float eps) {
  static_assert(D == 128, "half-wave rows need D == 128");
  __shared__ float red[2][8][D];
  const int li = threadIdx.x & 31, hw = threadIdx.x >> 5;
  const float4 av = reinterpret_cast<const float4*>(a)[li];
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
  for (long row = (long)blockIdx.x * 8 + hw; row < rows; row += (long)gridDim.x * 8) {
    float4 xv = reinterpret_cast<const float4*>(x + row * D)[li];
    float4 g = reinterpret_cast<const float4*>(dy + row * D)[li];
    if (res) {
      float4 r = reinterpret_cast<const float4*>(res + row * D)[li];
      xv.x += r.x; xv.y += r.y; xv.z += r.z; xv.w += r.w;
    }
    float mean = save_mean[row], rinv = save_rinv[row];
    float sd = 1.f / rinv - eps;
    float4 u = make_float4(xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean);
    pa.x += g.x * u.x * rinv; pa.y += g.y * u.y * rinv; pa.z += g.z * u.z * rinv; pa.w += g.w * u.w * rinv;
    pb.x += g.x; pb.y += g.y; pb.z += g.z; pb.w += g.w;
    float4 dn = make_float4(g.x * av.x, g.y * av.y, g.z * av.z, g.w * av.w);
    float s_dn = half_sum((dn.x + dn.y) + (dn.z + dn.w));
    float s_dnu = half_sum((dn.x * u.x + dn.y * u.y) + (dn.z * u.z + dn.w * u.w));
    float k = rinv * rinv * s_dnu / ((D - 1) * sd);
    float mdn = rinv * s_dn / D;
    float4 o;
    o.x = dn.x * rinv - k * u.x - mdn;
    o.y = dn.y * rinv - k * u.y - mdn;
    o.z = dn.z * rinv - k * u.z - mdn;
    o.w = dn.w * rinv - k * u.w - mdn;
    reinterpret_cast<float4*>(dx + row * D)[li] = o;
  }
  reinterpret_cast<float4*>(&red[0][hw][0])[li] = pa;
  reinterpret_cast<float4*>(&red[1][hw][0])[li] = pb;
  __syncthreads();
  {
    const int which = threadIdx.x >> 7, i = threadIdx.x & 127;      // 256 threads = 2 x D outputs
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < 8; ++h) s += red[which][h][i];
    atomicAdd(which ? &db[i] : &da[i], s);
  }
}

// ---------------------------------------------------------------------------------------
// PReLU with one shared slope
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ slope,
                                                        float* __restrict__ y, long n4) {
  const float a = slope[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x = v.x >= 0.f ? v.x : a * v.x;
    v.y = v.y >= 0.f ? v.y : a * v.y;
    v.z = v.z >= 0.f ? v.z : a * v.z;
    v.w = v.w >= 0.f ? v.w : a * v.w;
    reinterpret_cast<float4*>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy,
                                                        const float* __restrict__ x,
                                                        const float* __restrict__ slope,
                                                        float* __restrict__ dx, float* __restrict__ dslope,
                                                        long n4) {
  __shared__ float red[4];
  const float a = slope[0];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 d = reinterpret_cast<const float4*>(dy)[i];
    float4 o;
    o.x = v.x >= 0.f ? d.x : a * d.x;  acc += v.x >= 0.f ? 0.f : d.x * v.x;
    o.y = v.y >= 0.f ? d.y : a * d.y;  acc += v.y >= 0.f ? 0.f : d.y * v.y;
    o.z = v.z >= 0.f ? d.z : a * d.z;  acc += v.z >= 0.f ? 0.f : d.z * v.z;
    o.w = v.w >= 0.f ? d.w : a * d.w;  acc += v.w >= 0.f ? 0.f : d.w * v.w;
    reinterpret_cast<float4*>(dx)[i] = o;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dslope, red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------------------------------
// pixel-shuffle(2) + mish:  pre [N,H,W,4C] -> z [N,2H,2W,C];  thread = (n,h,w,c)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pshuf_mish_fwd_kernel(const float* __restrict__ pre,
                                                             float* __restrict__ z, long total, int H,
                                                             int W, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long p = i / C;
    int w = (int)(p % W);
    long t = p / W;
    int h = (int)(t % H);
    long n = t / H;
    float4 v = reinterpret_cast<const float4*>(pre)[i];
    size_t o = (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    z[o] = mish_f(v.x);
    z[o + C] = mish_f(v.y);
    z[o + (size_t)2 * W * C] = mish_f(v.z);
    z[o + (size_t)2 * W * C + C] = mish_f(v.w);
  }
}
__global__ __launch_bounds__(256) void pshuf_mish_bwd_kernel(const float* __restrict__ dz,
                                                             const float* __restrict__ pre,
                                                             float* __restrict__ dpre, long total, int H,
                                                             int W, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long p = i / C;
    int w = (int)(p % W);
    long t = p / W;
    int h = (int)(t % H);
    long n = t / H;
    float4 v = reinterpret_cast<const float4*>(pre)[i];
    size_t o = (((size_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    float4 g;
    g.x = dz[o] * mish_grad_f(v.x);
    g.y = dz[o + C] * mish_grad_f(v.y);
    g.z = dz[o + (size_t)2 * W * C] * mish_grad_f(v.z);
    g.w = dz[o + (size_t)2 * W * C + C] * mish_grad_f(v.w);
    reinterpret_cast<float4*>(dpre)[i] = g;
  }
}

// ---------------------------------------------------------------------------------------
// layout / small elementwise
// ---------------------------------------------------------------------------------------
// y_nhwc[n,p,c] = x_nchw[n,c,p]
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C,
                                    int HW) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long t = i / C;
    int p = (int)(t % HW);
    long n = t / HW;
    y[i] = x[((size_t)n * C + c) * HW + p];
  }
}
// y_nchw[n,c,p] = f(x_nhwc[n,p,c]),  f = tanh if do_tanh
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int C,
                                    int HW, int do_tanh) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int p = (int)(i % HW);
    long t = i / HW;
    int c = (int)(t % C);
    long n = t / C;
    float v = x[((size_t)n * HW + p) * C + c];
    y[i] = do_tanh ? tanhf(v) : v;
  }
}
// dx_nhwc[n,p,c] = dy_nchw[n,c,p] * (1 - y_nchw^2)
__global__ void tanh_bwd_to_nhwc_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                        float* __restrict__ dx, long total, int C, int HW) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long t = i / C;
    int p = (int)(t % HW);
    long n = t / HW;
    size_t j = ((size_t)n * C + c) * HW + p;
    float yy = y[j];
    dx[i] = dy[j] * (1.f - yy * yy);
  }
}

// tok[r, 0:Cf] = feat[r,:], tok[r, Cf:Cf+Cp] = pe[r % T, :]      (float4 granularity)
__global__ __launch_bounds__(256) void concat_pe_kernel(const float* __restrict__ feat,
                                                        const float* __restrict__ pe,
                                                        float* __restrict__ tok, long total4, int Cf,
                                                        int Cp, int T) {
  const int D4 = (Cf + Cp) / 4, Cf4 = Cf / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    int c4 = (int)(i % D4);
    long r = i / D4;
    float4 v;
    if (c4 < Cf4) v = reinterpret_cast<const float4*>(feat)[r * Cf4 + c4];
    else v = reinterpret_cast<const float4*>(pe)[(r % T) * (Cp / 4) + (c4 - Cf4)];
    reinterpret_cast<float4*>(tok)[i] = v;
  }
}
// out[r, 0:w] = x[r, c0:c0+w]  (+ add[r,0:w] if add)      all multiples of 4
__global__ __launch_bounds__(256) void slice_cols_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ add,
                                                         float* __restrict__ out, long total4, int ld,
                                                         int c0, int w) {
  const int w4 = w / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    int c4 = (int)(i % w4);
    long r = i / w4;
    float4 v = *reinterpret_cast<const float4*>(x + r * ld + c0 + c4 * 4);
    if (add) {
      float4 a = reinterpret_cast<const float4*>(add)[i];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// y = keep ? x / (1-p) : 0, mask from rng_hash(seed, element index)
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      long n, float p, uint64_t seed,
                                                      const uint64_t* __restrict__ epoch) {
  seed = focr_epoch_seed(seed, epoch);
  const uint32_t thr = (uint32_t)(p * 4294967296.0);
  const float ik = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = rng_hash(seed, (uint64_t)i) >= thr ? x[i] * ik : 0.f;
}

// out[0] += scale * sum (a-b)^2
__global__ __launch_bounds__(256) void mse_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float d = a[i] - b[i];
    acc += d * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * scale);
}
// da = upstream[0] * 2 (a-b) / n
__global__ __launch_bounds__(256) void mse_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ upstream,
                                                      float* __restrict__ da, long n) {
  const float k = upstream[0] * 2.f / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    da[i] = k * (a[i] - b[i]);
}
// y = alpha * x + (add ? add : 0)   (generic axpby used for small glue)
__global__ __launch_bounds__(256) void axpy_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                   float* __restrict__ y, long n, float alpha) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = alpha * x[i] + (add ? add[i] : 0.f);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static inline int ew_grid(long n) {
  long g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
static inline int row_slabs(long rows) {
  long s = (rows + 255) / 256;
  if (s > 512) s = 512;
  if (s < 1) s = 1;
  return (int)s;
}
#define MEMSET0(ptr, bytes)                                                   \
  do {                                                                        \
    if (hipMemsetAsync((ptr), 0, (bytes), stream) != hipSuccess) {            \
      focr_set_error("%s: memset failed", __func__);                          \
      return FOCR_EHIP;                                                       \
    }                                                                         \
  } while (0)

// ws: focr_bn_ws_floats(rows, C) floats of scratch.  running_mean/var/nbt may be null (no running update).
// Statistics are reduced deterministically (slab partials + fixed-order fold): bit-identical run to run.

// ---------------------------------------------------------------------------------------
// Small tensors (the STN head's BatchNorm layers on 2x8 .. 1x2 maps and its BatchNorm1d, stn_head.py:34-47: rows <= 2048;
// measured on the 8192 x 128 layer: 50 us forward / 74 us backward against 32 / 22 us for the many-block form -- a block's
// strided walk over 32+ rows per thread is slower than the launches it saves):
// statistics, finalize and apply in ONE launch, and reduce + apply of the backward in one launch.  The five-launch form
// costs ~40 us per layer there (8-block column reductions of 64 dependent loads per thread, then three launches that do
// almost nothing), and the head is a chain of ~60 such launches on the step's critical path.  One block per float4
// column: it owns its four channels for all rows, so mean -> centred variance -> apply need no grid-wide step; the
// re-reads hit L2 (the whole tensor is <= 8 MB).  Same formulas and the same two-pass variance as the large form,
// fixed-order block reductions: deterministic.
// ---------------------------------------------------------------------------------------
#define BN_SMALL_MAX_ROWS 2048
__device__ __forceinline__ float4 bn_block_sum4(float4 v, float4* red) {
  __syncthreads();                                   // (the previous use of `red` is over)
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      float4 a = red[threadIdx.x], b = red[threadIdx.x + w];
      red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    __syncthreads();
  }
  return red[0];
}
// (eight rows per thread and iteration: the loads of a pass are in flight together)
#define BN_SMALL_U 8
#define BN_SMALL_LOAD(dst, base, pitch)                                                                       \
  _Pragma("unroll") for (int u = 0; u < BN_SMALL_U; ++u) {                                                    \
    const int r_ = r0 + 256 * u;                                                                              \
    dst[u] = r_ < rows ? *reinterpret_cast<const float4*>((base) + (size_t)r_ * (pitch)) : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
__global__ __launch_bounds__(256) void bn_small_train_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ rmean,
    float* __restrict__ rvar, long long* nbt, const float* __restrict__ res, float* __restrict__ y,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, int rows, int C, float momentum, float eps, int act) {
  __shared__ float4 red[256];
  const int c = blockIdx.x * 4;
  const float* xc = x + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = threadIdx.x; r0 < rows; r0 += 256 * BN_SMALL_U) {
    float4 v[BN_SMALL_U];
    BN_SMALL_LOAD(v, xc, C)
#pragma unroll
    for (int u = 0; u < BN_SMALL_U; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  s = bn_block_sum4(s, red);
  const float4 mu = make_float4(s.x / (float)rows, s.y / (float)rows, s.z / (float)rows, s.w / (float)rows);
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r0 = threadIdx.x; r0 < rows; r0 += 256 * BN_SMALL_U) {
    float4 v[BN_SMALL_U];
    BN_SMALL_LOAD(v, xc, C)
#pragma unroll
    for (int u = 0; u < BN_SMALL_U; ++u) {
      if (r0 + 256 * u < rows) {
        const float dx = v[u].x - mu.x, dy = v[u].y - mu.y, dz = v[u].z - mu.z, dw = v[u].w - mu.w;
        q.x += dx * dx; q.y += dy * dy; q.z += dz * dz; q.w += dw * dw;
      }
    }
  }
  q = bn_block_sum4(q, red);
  const float4 var = make_float4(q.x / (float)rows, q.y / (float)rows, q.z / (float)rows, q.w / (float)rows);
  const float4 is = make_float4(rsqrtf(var.x + eps), rsqrtf(var.y + eps), rsqrtf(var.z + eps), rsqrtf(var.w + eps));
  if (threadIdx.x == 0) {
    *reinterpret_cast<float4*>(save_mean + c) = mu;
    *reinterpret_cast<float4*>(save_invstd + c) = is;
    if (rmean) {
      const float k = rows > 1 ? (float)rows / (float)(rows - 1) : 1.f;
      const float m_[4] = {mu.x, mu.y, mu.z, mu.w}, v_[4] = {var.x, var.y, var.z, var.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float unb = rows > 1 ? v_[e] * k : v_[e];
        rmean[c + e] = (1.f - momentum) * rmean[c + e] + momentum * m_[e];
        rvar[c + e] = (1.f - momentum) * rvar[c + e] + momentum * unb;
      }
    }
    if (c == 0 && nbt) *nbt += 1;
  }
  const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
  for (int r0 = threadIdx.x; r0 < rows; r0 += 256 * BN_SMALL_U) {
    float4 v[BN_SMALL_U], rr[BN_SMALL_U];
    BN_SMALL_LOAD(v, xc, C)
    if (res) { BN_SMALL_LOAD(rr, res + c, C) }
#pragma unroll
    for (int u = 0; u < BN_SMALL_U; ++u) {
      const int r = r0 + 256 * u;
      if (r < rows) {
        float4 o;
        o.x = act_fwd(g.x * (v[u].x - mu.x) * is.x + b.x, act);
        o.y = act_fwd(g.y * (v[u].y - mu.y) * is.y + b.y, act);
        o.z = act_fwd(g.z * (v[u].z - mu.z) * is.z + b.z, act);
        o.w = act_fwd(g.w * (v[u].w - mu.w) * is.w + b.w, act);
        if (res) { o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w; }
        *reinterpret_cast<float4*>(y + c + (size_t)r * C) = o;
      }
    }
  }
}
// backward: sum g, sum g * xhat over the rows (g = dz * act'(gamma * xhat + beta)), then
// dx = gamma * invstd * (g - sum_g / rows - xhat * sum_gx / rows); dbeta = sum g, dgamma = sum g * xhat
__global__ __launch_bounds__(256) void bn_small_train_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, int act, int lddz) {
  __shared__ float4 red[256];
  const int c = blockIdx.x * 4;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + c), b4 = *reinterpret_cast<const float4*>(beta + c);
  const float4 m4 = *reinterpret_cast<const float4*>(mean + c), i4 = *reinterpret_cast<const float4*>(invstd + c);
  const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
  const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ii[4] = {i4.x, i4.y, i4.z, i4.w};
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r0 = threadIdx.x; r0 < rows; r0 += 256 * BN_SMALL_U) {
    float4 xv[BN_SMALL_U], dv[BN_SMALL_U];
    BN_SMALL_LOAD(xv, x + c, C)
    BN_SMALL_LOAD(dv, dz + c, lddz)
#pragma unroll
    for (int u = 0; u < BN_SMALL_U; ++u) {
      const float xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {                        // (rows past the end were loaded as zeros: dz = 0 -> g = 0)
        const float xh = (xx[e] - mm[e]) * ii[e];
        const float g = dd[e] * act_grad(gg[e] * xh + bb[e], act);
        s0[e] += g;
        s1[e] += g * xh;
      }
    }
  }
  const float4 sg = bn_block_sum4(make_float4(s0[0], s0[1], s0[2], s0[3]), red);
  const float4 sgx = bn_block_sum4(make_float4(s1[0], s1[1], s1[2], s1[3]), red);
  if (threadIdx.x == 0) {
    *reinterpret_cast<float4*>(dbeta + c) = sg;
    *reinterpret_cast<float4*>(dgamma + c) = sgx;
  }
  const float inv_rows = 1.f / (float)rows;
  const float a0[4] = {sg.x * inv_rows, sg.y * inv_rows, sg.z * inv_rows, sg.w * inv_rows};
  const float a1[4] = {sgx.x * inv_rows, sgx.y * inv_rows, sgx.z * inv_rows, sgx.w * inv_rows};
  for (int r0 = threadIdx.x; r0 < rows; r0 += 256 * BN_SMALL_U) {
    float4 xv[BN_SMALL_U], dv[BN_SMALL_U];
    BN_SMALL_LOAD(xv, x + c, C)
    BN_SMALL_LOAD(dv, dz + c, lddz)
#pragma unroll
    for (int u = 0; u < BN_SMALL_U; ++u) {
      const int r = r0 + 256 * u;
      if (r < rows) {
        const float xx[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, dd[4] = {dv[u].x, dv[u].y, dv[u].z, dv[u].w};
        float oo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xx[e] - mm[e]) * ii[e];
          const float g = dd[e] * act_grad(gg[e] * xh + bb[e], act);
          oo[e] = gg[e] * ii[e] * (g - a0[e] - xh * a1[e]);
        }
        *reinterpret_cast<float4*>(dx + c + (size_t)r * C) = make_float4(oo[0], oo[1], oo[2], oo[3]);
      }
    }
  }
}
#ifndef BN_NO_SMALL
// (rows AND elements: a block walks its four channels through all rows with a stride of C floats -- with many channels
// the many-block form is faster: the SLD recognizer's 8192 x 512 .. 1024 maps went from 17.1 to 19.9 ms per step on it)
#define BN_SMALL_MAX_ELEMS (1l << 20)
static inline bool bn_small(long rows, int C) { return rows <= BN_SMALL_MAX_ROWS && rows * C <= BN_SMALL_MAX_ELEMS; }
#else
static inline bool bn_small(long, int) { return false; }
#endif
extern "C" long focr_bn_ws_floats(long rows, int C) { return (long)(row_slabs(rows) + 2) * C; }

extern "C" int focr_bn_train_fwd(const float* x, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, long long* nbt,
                                 const float* residual, float* y, float* save_mean,
                                 float* save_invstd, float* ws, long rows, int C, float momentum,
                                 float eps, int act, hipStream_t stream) {
  FOCR_CHECK_ARG(x && gamma && beta && y && save_mean && save_invstd && ws, "null pointer");
  FOCR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "need C % 4 == 0");
  if (bn_small(rows, C)) {
    hipLaunchKernelGGL(bn_small_train_fwd_kernel, dim3(C / 4), 256, 0, stream, x, gamma, beta, running_mean, running_var,
                       nbt, residual, y, save_mean, save_invstd, (int)rows, C, momentum, eps, act);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  const int slabs = row_slabs(rows);
  float* sum = ws;                 // [C]
  float* sq = ws + C;              // [C]
  float* part = ws + 2 * C;        // [slabs][C]
  dim3 g(cdiv(C, 1024), slabs);
  hipLaunchKernelGGL(bn_colstat_kernel, g, 256, 0, stream, x, (const float*)nullptr, part, rows, C, 0);
  hipLaunchKernelGGL(bn_fold_kernel, dim3(C / 4), 256, 0, stream, (const float*)part, sum, slabs, C);
  hipLaunchKernelGGL(bn_colstat_kernel, g, 256, 0, stream, x, (const float*)sum, part, rows, C, 1);
  hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3(C / 4), 256, 0, stream, (const float*)part, (const float*)sum, sq,
                     save_mean, save_invstd, running_mean, running_var, nbt, rows, slabs, C, momentum, eps);
  long total4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, x, gamma, beta,
                     (const float*)save_mean, (const float*)save_invstd, residual, y, total4, C, act);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// Batch statistics from per-tile partial sums produced by the convolution that wrote x (focr_conv3x3_frag_fwd's
// `stats` output: part[tile][C][2] = (sum, sum of squares) of that tile's outputs): no pass over x for the statistics.
// One block per channel; every thread adds its tiles in double precision, then a fixed-order LDS tree: deterministic.
// var = E[x^2] - mean^2 evaluated in double on fp32 partials of <= 128 values each.
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int nparts,
                                                                float* __restrict__ save_mean,
                                                                float* __restrict__ save_invstd,
                                                                float* __restrict__ rmean, float* __restrict__ rvar,
                                                                long long* nbt, long rows, int C, float momentum,
                                                                float eps) {
  __shared__ double r1[256], r2[256];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int j = threadIdx.x; j < nparts; j += 256) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)j * C + c) * 2);
    s1 += (double)v.x;
    s2 += (double)v.y;
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      r1[threadIdx.x] += r1[threadIdx.x + w];
      r2[threadIdx.x] += r2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = r1[0] / (double)rows;
    double var = r2[0] / (double)rows - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      const float unb = (float)(rows > 1 ? var * ((double)rows / (double)(rows - 1)) : var);
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
    if (c == 0 && nbt) *nbt += 1;
  }
}

extern "C" int focr_bn_train_fwd_stats(const float* x, const float* part, int nparts, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, long long* nbt,
                                       const float* residual, float* y, float* save_mean, float* save_invstd,
                                       long rows, int C, float momentum, float eps, int act, hipStream_t stream) {
  FOCR_CHECK_ARG(part && gamma && beta && save_mean && save_invstd && (x || !y), "null pointer");
  FOCR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && nparts > 0, "need C % 4 == 0, nparts > 0");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(C), 256, 0, stream, part, nparts, save_mean, save_invstd,
                     running_mean, running_var, nbt, rows, C, momentum, eps);
  if (!y) {                 // statistics only: the consumer normalises on load (focr_fe_qkv_fwd_bn)
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  long total4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, x, gamma, beta,
                     (const float*)save_mean, (const float*)save_invstd, residual, y, total4, C, act);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// eval: normalise with running stats; invstd_out: C floats (kept for the backward)
extern "C" int focr_bn_eval_fwd(const float* x, const float* gamma, const float* beta,
                                const float* running_mean, const float* running_var,
                                const float* residual, float* y, float* invstd_out, long rows, int C,
                                float eps, int act, hipStream_t stream) {
  FOCR_CHECK_ARG(x && gamma && beta && running_mean && running_var && y && invstd_out, "null pointer");
  FOCR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "need C % 4 == 0");
  hipLaunchKernelGGL(bn_eval_prep_kernel, dim3(cdiv(C, 64)), 64, 0, stream, running_var, invstd_out, C, eps);
  long total4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, x, gamma, beta, running_mean,
                     (const float*)invstd_out, residual, y, total4, C, act);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// eval with the inverse standard deviation already known (focr_bn_eval_fwd's invstd_out of an earlier call with the same
// running_var and eps: a frozen recognizer normalises with the same statistics every step -- one launch instead of two)
extern "C" int focr_bn_eval_apply(const float* x, const float* gamma, const float* beta, const float* running_mean,
                                  const float* invstd, const float* residual, float* y, long rows, int C, int act,
                                  hipStream_t stream) {
  FOCR_CHECK_ARG(x && gamma && beta && running_mean && invstd && y, "null pointer");
  FOCR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "need C % 4 == 0");
  long total4 = rows * C / 4;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, x, gamma, beta, running_mean, invstd, residual,
                     y, total4, C, act);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// train-mode backward.  dgamma/dbeta: C floats each (overwritten); ws: focr_bn_bwd_ws_floats(rows, C) floats.
// The two channel reductions (sum g*xhat = dgamma, sum g = dbeta) are slab partials + a fixed-order fold.
// train == 0: eval-mode backward (mean = running_mean, no batch-statistics terms, no dgamma/dbeta, ws unused).
static inline int bwd_slabs(long rows) {
  long s = (rows + 63) / 64;
#ifndef BN_BWD_MAXSLABS
#define BN_BWD_MAXSLABS 2048
#endif
  // wide, short tensors (the SLD / text-focus ResNets: 8192 rows x 512 channels): 64-row slabs are 128 blocks, half the
  // chip and a 26 us launch for 33 MB (profiles/r06b_c5_bygrid.txt); 16-row slabs until there are 512 blocks
  if (s < 512) {
    long t = (rows + 15) / 16;
    if (t > 512) t = 512;
    if (t > s) s = t;
  }
  if (s > BN_BWD_MAXSLABS) s = BN_BWD_MAXSLABS;
  if (s < 1) s = 1;
  return (int)s;
}
extern "C" long focr_bn_bwd_ws_floats(long rows, int C) { return (long)bwd_slabs(rows) * 2 * C; }

extern "C" int focr_bn_bwd(const float* dz, const float* x, const float* gamma, const float* beta,
                           const float* mean, const float* invstd, float* dx, float* dgamma,
                           float* dbeta, float* ws, long rows, int C, int act, int train, int lddz,
                           hipStream_t stream) {
  if (lddz <= 0) lddz = C;
  FOCR_CHECK_ARG(lddz >= C && lddz % 4 == 0, "dz row pitch must be >= C and a multiple of 4");
  FOCR_CHECK_ARG(dz && x && gamma && beta && mean && invstd && dx, "null pointer");
  FOCR_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "need C % 4 == 0");
  long total4 = rows * C / 4;
  if (train && bn_small(rows, C)) {
    FOCR_CHECK_ARG(dgamma && dbeta, "null pointer");
    hipLaunchKernelGGL(bn_small_train_bwd_kernel, dim3(C / 4), 256, 0, stream, dz, x, gamma, beta, mean, invstd, dx, dgamma,
                       dbeta, (int)rows, C, act, lddz);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (train) {
    FOCR_CHECK_ARG(dgamma && dbeta && ws, "null pointer");
    const int slabs = bwd_slabs(rows);
    float* pg = ws;                       // [slabs][C]
    float* pgx = ws + (size_t)slabs * C;  // [slabs][C]
    dim3 g(cdiv(C, 1024), slabs);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, g, 256, 0, stream, dz, x, gamma, beta, mean, invstd, pg, pgx, rows, C, act, lddz);
    hipLaunchKernelGGL(bn_fold2_kernel, dim3(C / 4, 2), 256, 0, stream, (const float*)pg, dbeta, (const float*)pgx,
                       dgamma, slabs, C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, dz, x, gamma, beta, mean,
                       invstd, (const float*)dbeta, (const float*)dgamma, dx, total4, rows, C, act, lddz);
  } else {
    if (dgamma && dbeta && ws) {        // eval-mode statistics, trainable affine: the same two sums, no mean terms in dx
      const int slabs = bwd_slabs(rows);
      float* pg = ws;
      float* pgx = ws + (size_t)slabs * C;
      dim3 g(cdiv(C, 1024), slabs);
      hipLaunchKernelGGL(bn_bwd_reduce_kernel, g, 256, 0, stream, dz, x, gamma, beta, mean, invstd, pg, pgx, rows, C, act, lddz);
      hipLaunchKernelGGL(bn_fold2_kernel, dim3(C / 4, 2), 256, 0, stream, (const float*)pg, dbeta, (const float*)pgx,
                         dgamma, slabs, C);
    }
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(total4)), 256, 0, stream, dz, x, gamma, beta, mean,
                       invstd, (const float*)nullptr, (const float*)nullptr, dx, total4, rows, C, act, lddz);
  }
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

int focr_ln_any_fwd(const float* x, const float* res, const float* a, const float* b, float* y, float* save_mean,
                    float* save_rinv, long rows, int D, float eps, hipStream_t stream);
int focr_ln_any_bwd(const float* dy, const float* x, const float* res, const float* a, const float* save_mean,
                    const float* save_rinv, float* dx, float* da, float* db, long rows, int D, float eps,
                    hipStream_t stream);

extern "C" int focr_layernorm_fwd(const float* x, const float* residual, const float* a,
                                  const float* b, float* y, float* save_mean, float* save_rinv,
                                  long rows, int D, float eps, hipStream_t stream) {
  FOCR_CHECK_ARG(x && a && b && y && save_mean && save_rinv, "null pointer");
  FOCR_CHECK_ARG(D >= 2 && rows > 0, "bad shape");
  if (D != 128) {                    // generic width (the SLD decoder's D = 1024): csrc/sld_ops.hip
    focr_ln_any_fwd(x, residual, a, b, y, save_mean, save_rinv, rows, D, eps, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  hipLaunchKernelGGL((ln_fwd_kernel<128>), dim3(cdiv(rows, 8)), 256, 0, stream, x, residual, a, b, y,
                     save_mean, save_rinv, rows, eps);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_layernorm_bwd(const float* dy, const float* x, const float* residual,
                                  const float* a, const float* save_mean, const float* save_rinv,
                                  float* dx, float* da, float* db, long rows, int D, float eps,
                                  int prezeroed, hipStream_t stream) {
  FOCR_CHECK_ARG(dy && x && a && save_mean && save_rinv && dx && da && db, "null pointer");
  FOCR_CHECK_ARG(D >= 2 && rows > 0, "bad shape");
  if (!prezeroed) {
    MEMSET0(da, sizeof(float) * D);
    MEMSET0(db, sizeof(float) * D);
  }
  if (D != 128) {
    focr_ln_any_bwd(dy, x, residual, a, save_mean, save_rinv, dx, da, db, rows, D, eps, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
#ifndef LN_BWD_BLOCKS
#define LN_BWD_BLOCKS 512   // same-address atomics of the a_2/b_2 gradients dominate beyond this (2048: 78 us, 512: 54 us)
#endif
  long g = cdiv(rows, 8);
  if (g > LN_BWD_BLOCKS) g = LN_BWD_BLOCKS;
  hipLaunchKernelGGL((ln_bwd_kernel<128>), dim3((int)g), 256, 0, stream, dy, x, residual, a, save_mean,
                     save_rinv, dx, da, db, rows, eps);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_prelu_fwd(const float* x, const float* slope, float* y, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(x && slope && y && n > 0 && n % 4 == 0, "need n % 4 == 0");
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3(ew_grid(n / 4)), 256, 0, stream, x, slope, y, n / 4);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx,
                              float* dslope, long n, int prezeroed, hipStream_t stream) {
  FOCR_CHECK_ARG(dy && x && slope && dx && dslope && n > 0 && n % 4 == 0, "need n % 4 == 0");
  if (!prezeroed) MEMSET0(dslope, sizeof(float));
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(ew_grid(n / 4)), 256, 0, stream, dy, x, slope, dx, dslope, n / 4);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// pre [N,H,W,4C] -> z [N,2H,2W,C]
extern "C" int focr_pixelshuffle_mish_fwd(const float* pre, float* z, int N, int H, int W, int C,
                                          hipStream_t stream) {
  FOCR_CHECK_ARG(pre && z && N > 0 && H > 0 && W > 0 && C > 0, "bad argument");
  long total = (long)N * H * W * C;
  hipLaunchKernelGGL(pshuf_mish_fwd_kernel, dim3(ew_grid(total)), 256, 0, stream, pre, z, total, H, W, C);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_pixelshuffle_mish_bwd(const float* dz, const float* pre, float* dpre, int N, int H,
                                          int W, int C, hipStream_t stream) {
  FOCR_CHECK_ARG(dz && pre && dpre && N > 0 && H > 0 && W > 0 && C > 0, "bad argument");
  long total = (long)N * H * W * C;
  hipLaunchKernelGGL(pshuf_mish_bwd_kernel, dim3(ew_grid(total)), 256, 0, stream, dz, pre, dpre, total, H, W, C);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, hipStream_t stream) {
  FOCR_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0, "bad argument");
  long total = (long)N * C * HW;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid(total)), 256, 0, stream, x, y, total, C, HW);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, int do_tanh,
                                 hipStream_t stream) {
  FOCR_CHECK_ARG(x && y && N > 0 && C > 0 && HW > 0, "bad argument");
  long total = (long)N * C * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid(total)), 256, 0, stream, x, y, total, C, HW, do_tanh);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_tanh_bwd_to_nhwc(const float* dy_nchw, const float* y_nchw, float* dx_nhwc, int N,
                                     int C, int HW, hipStream_t stream) {
  FOCR_CHECK_ARG(dy_nchw && y_nchw && dx_nhwc && N > 0 && C > 0 && HW > 0, "bad argument");
  long total = (long)N * C * HW;
  hipLaunchKernelGGL(tanh_bwd_to_nhwc_kernel, dim3(ew_grid(total)), 256, 0, stream, dy_nchw, y_nchw, dx_nhwc,
                     total, C, HW);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_concat_pe(const float* feat, const float* pe, float* tok, long rows, int Cf, int Cp,
                              int T, hipStream_t stream) {
  FOCR_CHECK_ARG(feat && pe && tok && rows > 0 && Cf % 4 == 0 && Cp % 4 == 0 && T > 0, "bad argument");
  long total4 = rows * (Cf + Cp) / 4;
  hipLaunchKernelGGL(concat_pe_kernel, dim3(ew_grid(total4)), 256, 0, stream, feat, pe, tok, total4, Cf, Cp, T);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_slice_cols(const float* x, const float* add, float* out, long rows, int ld, int c0,
                               int w, hipStream_t stream) {
  FOCR_CHECK_ARG(x && out && rows > 0 && ld % 4 == 0 && c0 % 4 == 0 && w % 4 == 0 && c0 + w <= ld, "bad argument");
  long total4 = rows * w / 4;
  hipLaunchKernelGGL(slice_cols_kernel, dim3(ew_grid(total4)), 256, 0, stream, x, add, out, total4, ld, c0, w);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_dropout(const float* x, float* y, long n, float p, uint64_t seed, hipStream_t stream) {
  FOCR_CHECK_ARG(x && y && n > 0 && p >= 0.f && p < 1.f, "bad argument");
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), 256, 0, stream, x, y, n, p, seed, focr_seed_epoch());
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// out (1 float, overwritten) = mean (a-b)^2
extern "C" int focr_mse_fwd(const float* a, const float* b, float* out, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && out && n > 0, "bad argument");
  MEMSET0(out, sizeof(float));
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(ew_grid(n) > 256 ? 256 : ew_grid(n)), 256, 0, stream, a, b, out, n,
                     1.f / (float)n);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_mse_bwd(const float* a, const float* b, const float* upstream, float* da, long n,
                            hipStream_t stream) {
  FOCR_CHECK_ARG(a && b && upstream && da && n > 0, "bad argument");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(ew_grid(n)), 256, 0, stream, a, b, upstream, da, n);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_axpy(const float* x, const float* add, float* y, long n, float alpha, hipStream_t stream) {
  FOCR_CHECK_ARG(x && y && n > 0, "bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n)), 256, 0, stream, x, add, y, n, alpha);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// dx = scale * dy * (y > 0): backward of relu followed by a fused dropout whose kept elements were scaled by `scale`
__global__ __launch_bounds__(256) void relu_bwd_scaled_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dx, long n, float scale) {
  const long n4 = n >> 2;                                   // 16-byte accesses; the (< 4 element) tail below
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i], v = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(dx)[i] = make_float4(v.x > 0.f ? g.x * scale : 0.f, v.y > 0.f ? g.y * scale : 0.f,
                                                   v.z > 0.f ? g.z * scale : 0.f, v.w > 0.f ? g.w * scale : 0.f);
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < n; i += blockDim.x) dx[i] = y[i] > 0.f ? dy[i] * scale : 0.f;
}
extern "C" int focr_relu_bwd_scaled(const float* dy, const float* y, float* dx, long n, float scale,
                                    hipStream_t stream) {
  FOCR_CHECK_ARG(dy && y && dx && n > 0, "bad argument");
  hipLaunchKernelGGL(relu_bwd_scaled_kernel, dim3(ew_grid(n / 4 + 1)), 256, 0, stream, dy, y, dx, n, scale);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_relu_bwd(const float* dy, const float* y, float* dx, long n, hipStream_t stream) {
  FOCR_CHECK_ARG(dy && y && dx && n > 0, "bad argument");
  hipLaunchKernelGGL(relu_bwd_scaled_kernel, dim3(ew_grid(n / 4 + 1)), 256, 0, stream, dy, y, dx, n, 1.f);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// Pooling and resampling kernels of the path (all fp32, channel-last unless noted):
//   MaxPool2d            model/stn_head.py:34-42, model/crnn/crnn.py:52-62  (-inf padding)
//   TPS warp             model/tps_spatial_transformer.py:97-111 + F.grid_sample (bilinear,
//                        zero padding, align_corners=False)
//   bicubic(32x128->32x100) + luma   interfaces/base.py:319-325 (A = -0.75, align_corners=False)
#include "focr_common.h"

// ---------------------------------------------------------------------------------------
// max pooling, NHWC, arbitrary window/stride/pad.  idx = window-local argmax (kh*KW+kw).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, long total, int H, int W,
                                                          int C, int OH, int OW, int kh, int kw, int sh,
                                                          int sw, int ph, int pw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long t = i / C;
    int ox = (int)(t % OW);
    t /= OW;
    int oy = (int)(t % OH);
    long n = t / OH;
    float best = -INFINITY;
    int bi = 0;
    for (int a = 0; a < kh; ++a) {
      int iy = oy * sh - ph + a;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int b = 0; b < kw; ++b) {
        int ix = ox * sw - pw + b;
        if ((unsigned)ix >= (unsigned)W) continue;
        float v = x[(((size_t)n * H + iy) * W + ix) * C + c];
        if (v > best) { best = v; bi = a * kw + b; }
      }
    }
    y[i] = best;
    idx[i] = (uint8_t)bi;
  }
}
// four channels per thread (C % 4 == 0): float4 loads / store, the four argmax bytes as one word.  Same rule as above (first
// strict maximum in (a, b) scan order).  The scalar form took 20 us on each of the recognizer's three pooling layers whatever
// their size (34 - 67 MB in): one element per thread, four divisions each.
__global__ __launch_bounds__(256) void maxpool_fwd_vec4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               uint8_t* __restrict__ idx, long total4, int H, int W, int C,
                                                               int OH, int OW, int kh, int kw, int sh, int sw, int ph,
                                                               int pw) {
  const int C4 = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const long n = t / OH;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t bi[4] = {0u, 0u, 0u, 0u};
    for (int a = 0; a < kh; ++a) {
      const int iy = oy * sh - ph + a;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int b = 0; b < kw; ++b) {
        const int ix = ox * sw - pw + b;
        if ((unsigned)ix >= (unsigned)W) continue;
        const float4 v4 = reinterpret_cast<const float4*>(x)[(((size_t)n * H + iy) * W + ix) * C4 + c4];
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (v[e] > best[e]) { best[e] = v[e]; bi[e] = (uint32_t)(a * kw + b); }
      }
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(best[0], best[1], best[2], best[3]);
    reinterpret_cast<uint32_t*>(idx)[i] = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
  }
}
// gather form: dx[n,iy,ix,c] = sum over windows (oy,ox) containing it whose argmax is it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, long total, int H, int W,
                                                          int C, int OH, int OW, int kh, int kw, int sh,
                                                          int sw, int ph, int pw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    long t = i / C;
    int ix = (int)(t % W);
    t /= W;
    int iy = (int)(t % H);
    long n = t / H;
    float acc = 0.f;
    for (int a = 0; a < kh; ++a) {
      int ny = iy + ph - a;
      if (ny < 0 || ny % sh) continue;
      int oy = ny / sh;
      if (oy >= OH) continue;
      for (int b = 0; b < kw; ++b) {
        int nx = ix + pw - b;
        if (nx < 0 || nx % sw) continue;
        int ox = nx / sw;
        if (ox >= OW) continue;
        size_t o = (((size_t)n * OH + oy) * OW + ox) * C + c;
        if (idx[o] == a * kw + b) acc += dy[o];
      }
    }
    dx[i] = acc;
  }
}
// one thread = 4 consecutive channels of one input pixel (C % 4 == 0: float4 / uchar4 accesses, 32-bit index math)
// ypool != nullptr: the pooled tensor was max(relu(.)): a window whose maximum is 0 passes no gradient (relu'(y <= 0) = 0),
// every other window's arg-max element has relu' = 1 -- the relu backward of the producing layer is applied HERE and its
// own pass over the (2-4x larger) un-pooled gradient disappears
__global__ __launch_bounds__(256) void maxpool_bwd_vec4_kernel(const float* __restrict__ dy,
                                                               const uint8_t* __restrict__ idx,
                                                               const float* __restrict__ ypool,
                                                               float* __restrict__ dx, long total4, int H, int W,
                                                               int C, int OH, int OW, int kh, int kw, int sh,
                                                               int sw, int ph, int pw) {
  const int C4 = C >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long t0 = i / C4;
    const int ix = (int)(t0 % W);
    const long t1 = t0 / W;
    const int iy = (int)(t1 % H);
    const long n = t1 / H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = 0; a < kh; ++a) {
      const int ny = iy + ph - a;
      if (ny < 0 || ny % sh) continue;
      const int oy = ny / sh;
      if (oy >= OH) continue;
      for (int b = 0; b < kw; ++b) {
        const int nx = ix + pw - b;
        if (nx < 0 || nx % sw) continue;
        const int ox = nx / sw;
        if (ox >= OW) continue;
        const size_t o = (((size_t)n * OH + oy) * OW + ox) * C4 + c4;
        const uchar4 k = reinterpret_cast<const uchar4*>(idx)[o];
        float4 g = reinterpret_cast<const float4*>(dy)[o];
        if (ypool) {
          const float4 yp = reinterpret_cast<const float4*>(ypool)[o];
          g.x = yp.x > 0.f ? g.x : 0.f; g.y = yp.y > 0.f ? g.y : 0.f; g.z = yp.z > 0.f ? g.z : 0.f; g.w = yp.w > 0.f ? g.w : 0.f;
        }
        const int tap = a * kw + b;
        acc.x += k.x == tap ? g.x : 0.f;
        acc.y += k.y == tap ? g.y : 0.f;
        acc.z += k.z == tap ? g.z : 0.f;
        acc.w += k.w == tap ? g.w : 0.f;
      }
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
  }
}


// ---------------------------------------------------------------------------------------
// TPS warp: one block per sample.  NP = H*W output pixels, NC control points (+3 affine rows)
// ---------------------------------------------------------------------------------------
#define TPS_MAXK 32
__global__ __launch_bounds__(256) void tps_fwd_kernel(const float* __restrict__ img,    // [B,H,W,C]
                                                      const float* __restrict__ ctrl,   // [B,NC,2]
                                                      const float* __restrict__ invk,   // [K,K], K=NC+3
                                                      const float* __restrict__ repr,   // [H*W,K]
                                                      float* __restrict__ out,          // [B,H,W,C]
                                                      float* __restrict__ src,          // [B,H*W,2]
                                                      int H, int W, int C, int NC) {
  __shared__ float map[TPS_MAXK][2];
  const int b = blockIdx.x, K = NC + 3;
  if (threadIdx.x < 2 * K) {
    int j = threadIdx.x >> 1, d = threadIdx.x & 1;
    float s = 0.f;
    for (int i = 0; i < NC; ++i) s += invk[j * K + i] * ctrl[((size_t)b * NC + i) * 2 + d];
    map[j][d] = s;
  }
  __syncthreads();
  const float* im = img + (size_t)b * H * W * C;
  for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
    float sx = 0.f, sy = 0.f;
    for (int j = 0; j < K; ++j) {
      float r = repr[(size_t)p * K + j];
      sx += r * map[j][0];
      sy += r * map[j][1];
    }
    src[((size_t)b * H * W + p) * 2] = sx;
    src[((size_t)b * H * W + p) * 2 + 1] = sy;
    float gx = fminf(fmaxf(sx, 0.f), 1.f) * 2.f - 1.f;
    float gy = fminf(fmaxf(sy, 0.f), 1.f) * 2.f - 1.f;
    float fx = ((gx + 1.f) * W - 1.f) * 0.5f, fy = ((gy + 1.f) * H - 1.f) * 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    int x0 = (int)x0f, y0 = (int)y0f;
    float tx = fx - x0f, ty = fy - y0f;
    bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    for (int c = 0; c < C; ++c) {
      float v00 = (vx0 && vy0) ? im[((size_t)y0 * W + x0) * C + c] : 0.f;
      float v01 = (vx1 && vy0) ? im[((size_t)y0 * W + x0 + 1) * C + c] : 0.f;
      float v10 = (vx0 && vy1) ? im[((size_t)(y0 + 1) * W + x0) * C + c] : 0.f;
      float v11 = (vx1 && vy1) ? im[((size_t)(y0 + 1) * W + x0 + 1) * C + c] : 0.f;
      out[((size_t)b * H * W + p) * C + c] =
          (v00 * (1.f - tx) + v01 * tx) * (1.f - ty) + (v10 * (1.f - tx) + v11 * tx) * ty;
    }
  }
}

// d ctrl only (the warped tensor is the input image: no parameter lives upstream of it)
__global__ __launch_bounds__(256) void tps_bwd_kernel(const float* __restrict__ dout,   // [B,H,W,C]
                                                      const float* __restrict__ img,
                                                      const float* __restrict__ src,
                                                      const float* __restrict__ invk,
                                                      const float* __restrict__ repr,
                                                      float* __restrict__ dctrl,        // [B,NC,2]
                                                      int H, int W, int C, int NC) {
  __shared__ float dmap[TPS_MAXK][2];
  __shared__ float red[4][TPS_MAXK][2];
  const int b = blockIdx.x, K = NC + 3;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* im = img + (size_t)b * H * W * C;
  float acc[TPS_MAXK][2];
#pragma unroll
  for (int j = 0; j < TPS_MAXK; ++j) { acc[j][0] = 0.f; acc[j][1] = 0.f; }
  for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
    float sx = src[((size_t)b * H * W + p) * 2], sy = src[((size_t)b * H * W + p) * 2 + 1];
    bool inx = sx >= 0.f && sx <= 1.f, iny = sy >= 0.f && sy <= 1.f;
    float gx = fminf(fmaxf(sx, 0.f), 1.f) * 2.f - 1.f;
    float gy = fminf(fmaxf(sy, 0.f), 1.f) * 2.f - 1.f;
    float fx = ((gx + 1.f) * W - 1.f) * 0.5f, fy = ((gy + 1.f) * H - 1.f) * 0.5f;
    float x0f = floorf(fx), y0f = floorf(fy);
    int x0 = (int)x0f, y0 = (int)y0f;
    float tx = fx - x0f, ty = fy - y0f;
    bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    float dfx = 0.f, dfy = 0.f;
    for (int c = 0; c < C; ++c) {
      float g = dout[((size_t)b * H * W + p) * C + c];
      float v00 = (vx0 && vy0) ? im[((size_t)y0 * W + x0) * C + c] : 0.f;
      float v01 = (vx1 && vy0) ? im[((size_t)y0 * W + x0 + 1) * C + c] : 0.f;
      float v10 = (vx0 && vy1) ? im[((size_t)(y0 + 1) * W + x0) * C + c] : 0.f;
      float v11 = (vx1 && vy1) ? im[((size_t)(y0 + 1) * W + x0 + 1) * C + c] : 0.f;
      dfx += g * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
      dfy += g * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
    }
    // fx = (gx+1)*W/2 - 0.5, gx = 2*clamp(sx) - 1   ->  d fx / d sx = W inside the clamp range
    float dsx = inx ? dfx * W : 0.f, dsy = iny ? dfy * H : 0.f;
#pragma unroll
    for (int j = 0; j < TPS_MAXK; ++j) {
      if (j < K) {
        float r = repr[(size_t)p * K + j];
        acc[j][0] += r * dsx;
        acc[j][1] += r * dsy;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TPS_MAXK; ++j) {
    float a0 = wave_sum(acc[j][0]), a1 = wave_sum(acc[j][1]);
    if (lane == 0) { red[wv][j][0] = a0; red[wv][j][1] = a1; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * K) {
    int j = threadIdx.x >> 1, d = threadIdx.x & 1;
    dmap[j][d] = red[0][j][d] + red[1][j][d] + red[2][j][d] + red[3][j][d];
  }
  __syncthreads();
  if (threadIdx.x < 2 * NC) {
    int i = threadIdx.x >> 1, d = threadIdx.x & 1;
    float s = 0.f;
    for (int j = 0; j < K; ++j) s += invk[j * K + i] * dmap[j][d];
    dctrl[((size_t)b * NC + i) * 2 + d] = s;
  }
}

// ---------------------------------------------------------------------------------------
// bicubic width resample (IW -> OW, height unchanged) + luma, NCHW in -> [B,H,OW] out
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_w(float t, float w[4]) {
  const float A = -0.75f;
  float t1 = t + 1.f, t2 = 1.f - t, t3 = 2.f - t;
  w[0] = ((A * t1 - 5.f * A) * t1 + 8.f * A) * t1 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * t2 - (A + 3.f)) * t2 * t2 + 1.f;
  w[3] = ((A * t3 - 5.f * A) * t3 + 8.f * A) * t3 - 4.f * A;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void bicubic_gray_fwd_kernel(const float* __restrict__ x,   // [B,Cx,H,IW]
                                                               float* __restrict__ y,         // [B,H,OW]
                                                               long total, int Cx, int H, int IW, int OW) {
  const float scale = (float)IW / (float)OW;
  const float cw[3] = {0.299f, 0.587f, 0.114f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ox = (int)(i % OW);
    long t = i / OW;
    int yy = (int)(t % H);
    long b = t / H;
    float sx = scale * (ox + 0.5f) - 0.5f;
    float x0f = floorf(sx);
    int x0 = (int)x0f;
    float w[4];
    cubic_w(sx - x0f, w);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* row = x + (((size_t)b * Cx + c) * H + yy) * IW;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += w[k] * row[clampi(x0 - 1 + k, 0, IW - 1)];
      acc += cw[c] * s;
    }
    y[i] = acc;
  }
}
// dx [B,Cx,H,IW] (channels >= 3 get zero), gather over the outputs that touch each input
__global__ __launch_bounds__(256) void bicubic_gray_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                               long total, int Cx, int H, int IW, int OW) {
  const float scale = (float)IW / (float)OW;
  const float cw[3] = {0.299f, 0.587f, 0.114f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int ix = (int)(i % IW);
    long t = i / IW;
    int yy = (int)(t % H);
    t /= H;
    int c = (int)(t % Cx);
    long b = t / Cx;
    float acc = 0.f;
    if (c < 3) {
      int lo = (int)floorf((ix - 3.f) / scale) - 1, hi = (int)ceilf((ix + 3.f) / scale) + 1;
      lo = lo < 0 ? 0 : lo;
      hi = hi > OW - 1 ? OW - 1 : hi;
      const float* drow = dy + ((size_t)b * H + yy) * OW;
      for (int ox = lo; ox <= hi; ++ox) {
        float sx = scale * (ox + 0.5f) - 0.5f;
        float x0f = floorf(sx);
        int x0 = (int)x0f;
        float w[4];
        cubic_w(sx - x0f, w);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (clampi(x0 - 1 + k, 0, IW - 1) == ix) s += w[k];
        acc += s * drow[ox];
      }
      acc *= cw[c];
    }
    dx[i] = acc;
  }
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

extern "C" int focr_maxpool_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, int kh,
                                int kw, int sh, int sw, int ph, int pw, hipStream_t stream) {
  FOCR_CHECK_ARG(x && y && idx, "null pointer");
  int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
  FOCR_CHECK_ARG(OH > 0 && OW > 0 && kh * kw <= 255, "bad geometry");
  long total = (long)N * OH * OW * C;
#ifndef MAXPOOL_NO_VEC4
  if (C % 4 == 0)
    hipLaunchKernelGGL(maxpool_fwd_vec4_kernel, dim3(ew_grid(total / 4)), 256, 0, stream, x, y, idx, total / 4, H, W, C, OH,
                       OW, kh, kw, sh, sw, ph, pw);
  else
#endif
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total)), 256, 0, stream, x, y, idx, total, H, W, C, OH, OW,
                     kh, kw, sh, sw, ph, pw);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
static int maxpool_bwd_impl(const float* dy, const uint8_t* idx, const float* ypool, float* dx, int N, int H, int W, int C,
                            int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t stream);
extern "C" int focr_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C,
                                int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t stream) {
  return maxpool_bwd_impl(dy, idx, nullptr, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, stream);
}
// maxpool backward fused with the backward of the relu that produced the pooled input (crnn.py:52-63: conv -> relu -> pool):
// ypool = the pooling layer's forward output; C % 4 == 0
extern "C" int focr_maxpool_relu_bwd(const float* dy, const uint8_t* idx, const float* ypool, float* dx, int N, int H,
                                     int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t stream) {
  FOCR_CHECK_ARG(ypool && C % 4 == 0, "needs the pooled forward output and C % 4 == 0");
  return maxpool_bwd_impl(dy, idx, ypool, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, stream);
}
static int maxpool_bwd_impl(const float* dy, const uint8_t* idx, const float* ypool, float* dx, int N, int H, int W, int C,
                            int kh, int kw, int sh, int sw, int ph, int pw, hipStream_t stream) {
  FOCR_CHECK_ARG(dy && dx && idx, "null pointer");
  int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
  FOCR_CHECK_ARG(OH > 0 && OW > 0, "bad geometry");
  long total = (long)N * H * W * C;
  if (C % 4 == 0)
    hipLaunchKernelGGL(maxpool_bwd_vec4_kernel, dim3(ew_grid(total / 4)), 256, 0, stream, dy, idx, ypool, dx, total / 4, H, W, C,
                       OH, OW, kh, kw, sh, sw, ph, pw);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total)), 256, 0, stream, dy, idx, dx, total, H, W, C, OH, OW,
                       kh, kw, sh, sw, ph, pw);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_tps_fwd(const float* img, const float* ctrl, const float* inv_kernel,
                            const float* coord_repr, float* out, float* src, int B, int H, int W, int C,
                            int NC, hipStream_t stream) {
  FOCR_CHECK_ARG(img && ctrl && inv_kernel && coord_repr && out && src, "null pointer");
  FOCR_CHECK_ARG(B > 0 && NC + 3 <= TPS_MAXK && 2 * (NC + 3) <= 256, "too many control points");
  hipLaunchKernelGGL(tps_fwd_kernel, dim3(B), 256, 0, stream, img, ctrl, inv_kernel, coord_repr, out, src, H, W, C, NC);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
// d loss / d image of the bilinear sampling (F.grid_sample backward w.r.t. its input, zeros padding,
// tps_spatial_transformer.py:10-18): every output pixel scatters its gradient to the four cells it read.
// dimg must be zeroed by the caller (focr_tps_bwd_img does it); fp32 atomics (the image has 3-4 channels).
__global__ __launch_bounds__(256) void tps_bwd_img_kernel(const float* __restrict__ dout,   // [B,H,W,C]
                                                          const float* __restrict__ src,    // [B,H*W,2]
                                                          float* __restrict__ dimg,         // [B,H,W,C]
                                                          int H, int W, int C) {
  const int b = blockIdx.x;
  float* di = dimg + (size_t)b * H * W * C;
  for (int p = threadIdx.x; p < H * W; p += blockDim.x) {
    const float sx = src[((size_t)b * H * W + p) * 2], sy = src[((size_t)b * H * W + p) * 2 + 1];
    const float gx = fminf(fmaxf(sx, 0.f), 1.f) * 2.f - 1.f;
    const float gy = fminf(fmaxf(sy, 0.f), 1.f) * 2.f - 1.f;
    const float fx = ((gx + 1.f) * W - 1.f) * 0.5f, fy = ((gy + 1.f) * H - 1.f) * 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float tx = fx - x0f, ty = fy - y0f;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)(x0 + 1) < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)(y0 + 1) < (unsigned)H;
    for (int c = 0; c < C; ++c) {
      const float g = dout[((size_t)b * H * W + p) * C + c];
      if (vx0 && vy0) atomicAdd(&di[((size_t)y0 * W + x0) * C + c], g * (1.f - tx) * (1.f - ty));
      if (vx1 && vy0) atomicAdd(&di[((size_t)y0 * W + x0 + 1) * C + c], g * tx * (1.f - ty));
      if (vx0 && vy1) atomicAdd(&di[((size_t)(y0 + 1) * W + x0) * C + c], g * (1.f - tx) * ty);
      if (vx1 && vy1) atomicAdd(&di[((size_t)(y0 + 1) * W + x0 + 1) * C + c], g * tx * ty);
    }
  }
}

extern "C" int focr_tps_bwd_img(const float* dout, const float* src, float* dimg, int B, int H, int W, int C,
                                hipStream_t stream) {
  FOCR_CHECK_ARG(dout && src && dimg && B > 0 && H > 0 && W > 0 && C > 0, "bad argument");
  if (hipMemsetAsync(dimg, 0, sizeof(float) * (size_t)B * H * W * C, stream) != hipSuccess) {
    focr_set_error("focr_tps_bwd_img: memset failed");
    return FOCR_EHIP;
  }
  hipLaunchKernelGGL(tps_bwd_img_kernel, dim3(B), 256, 0, stream, dout, src, dimg, H, W, C);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_tps_bwd(const float* dout, const float* img, const float* src, const float* inv_kernel,
                            const float* coord_repr, float* dctrl, int B, int H, int W, int C, int NC,
                            hipStream_t stream) {
  FOCR_CHECK_ARG(dout && img && src && inv_kernel && coord_repr && dctrl, "null pointer");
  FOCR_CHECK_ARG(B > 0 && NC + 3 <= TPS_MAXK && 2 * (NC + 3) <= 256, "too many control points");
  hipLaunchKernelGGL(tps_bwd_kernel, dim3(B), 256, 0, stream, dout, img, src, inv_kernel, coord_repr, dctrl, H, W, C, NC);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_bicubic_gray_fwd(const float* x_nchw, float* y, int B, int Cx, int H, int IW, int OW,
                                     hipStream_t stream) {
  FOCR_CHECK_ARG(x_nchw && y && B > 0 && Cx >= 3 && H > 0 && IW > 1 && OW > 0, "bad argument");
  long total = (long)B * H * OW;
  hipLaunchKernelGGL(bicubic_gray_fwd_kernel, dim3(ew_grid(total)), 256, 0, stream, x_nchw, y, total, Cx, H, IW, OW);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}
extern "C" int focr_bicubic_gray_bwd(const float* dy, float* dx_nchw, int B, int Cx, int H, int IW, int OW,
                                     hipStream_t stream) {
  FOCR_CHECK_ARG(dy && dx_nchw && B > 0 && Cx >= 3 && H > 0 && IW > 1 && OW > 0, "bad argument");
  long total = (long)B * Cx * H * IW;
  hipLaunchKernelGGL(bicubic_gray_bwd_kernel, dim3(ew_grid(total)), 256, 0, stream, dy, dx_nchw, total, Cx, H, IW, OW);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

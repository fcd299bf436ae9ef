// Evaluation metrics of the harness on the device: PSNR's squared-error sum and the SSIM map sum
// in ONE pass over the SR / HR pair, reference definitions (utils/ssim_psnr.py:9-15, 18-78):
//   PSNR: mse over the first 3 channels of (img1*255 - img2*255)^2;
//   SSIM: 11x11 Gaussian window (sigma 1.5, separable weights g[i]*g[j] formed in fp32 like the reference's mm),
//         zero padding 5, per-channel (depthwise) local means / variances / covariance, C1 = 0.01^2, C2 = 0.03^2.
// Inputs are NCHW fp32 (the harness evaluates the model's NCHW outputs), C >= 3, only the first three channels count.
// Per-image sums are reduced in a fixed order (block partials + one fold block per image): bit-identical run to run.
#include "focr_common.h"

#define EM_WMAX 15
struct EvalWin { float g[EM_WMAX]; int size; };

__global__ __launch_bounds__(256) void psnr_ssim_kernel(const float* __restrict__ A, const float* __restrict__ Bm,
                                                        float* __restrict__ part,   // [img][blocks][2]
                                                        int C, int H, int W, EvalWin win) {
  __shared__ float r1[256], r2[256];
  const int img = blockIdx.y;
  const int per = 3 * H * W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float sq = 0.f, ss = 0.f;
  if (i < per) {
    const int c = i / (H * W), rem = i - c * H * W, y = rem / W, x = rem - y * W;
    const float* a = A + ((size_t)img * C + c) * H * W;
    const float* b = Bm + ((size_t)img * C + c) * H * W;
    const float d = a[y * W + x] * 255.f - b[y * W + x] * 255.f;
    sq = d * d;
    const int half = win.size / 2;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
    for (int dy = 0; dy < win.size; ++dy) {
      const int yy = y + dy - half;
      if ((unsigned)yy >= (unsigned)H) continue;
      for (int dx = 0; dx < win.size; ++dx) {
        const int xx = x + dx - half;
        if ((unsigned)xx >= (unsigned)W) continue;
        const float w = win.g[dy] * win.g[dx];
        const float p = a[yy * W + xx], q = b[yy * W + xx];
        mu1 += w * p;
        mu2 += w * q;
        e11 += w * (p * p);
        e22 += w * (q * q);
        e12 += w * (p * q);
      }
    }
    const float m11 = mu1 * mu1, m22 = mu2 * mu2, m12 = mu1 * mu2;
    const float s11 = e11 - m11, s22 = e22 - m22, s12 = e12 - m12;
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    ss = ((2.f * m12 + c1) * (2.f * s12 + c2)) / ((m11 + m22 + c1) * (s11 + s22 + c2));
  }
  r1[threadIdx.x] = sq;
  r2[threadIdx.x] = ss;
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
    part[((size_t)img * gridDim.x + blockIdx.x) * 2] = r1[0];
    part[((size_t)img * gridDim.x + blockIdx.x) * 2 + 1] = r2[0];
  }
}

__global__ __launch_bounds__(64) void psnr_ssim_fold_kernel(const float* __restrict__ part, float* __restrict__ sqsum,
                                                            float* __restrict__ ssimsum, int nblocks) {
  const int img = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (int j = threadIdx.x; j < nblocks; j += 64) {
    a += (double)part[((size_t)img * nblocks + j) * 2];
    b += (double)part[((size_t)img * nblocks + j) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    b += __shfl_xor(b, o, 64);
  }
  if (threadIdx.x == 0) {
    sqsum[img] = (float)a;
    ssimsum[img] = (float)b;
  }
}

extern "C" long focr_psnr_ssim_ws_floats(int B, int H, int W) { return (long)B * cdiv(3L * H * W, 256) * 2; }

// window: `window_size` fp32 taps (the reference's normalised 1-D Gaussian).  sq_sum[b] = sum over the 3 x H x W
// elements of (255 a - 255 b)^2, ssim_sum[b] = sum of the SSIM map; the caller divides (means, log10).
extern "C" int focr_psnr_ssim(const float* img1, const float* img2, const float* window_host, int window_size,
                              float* sq_sum, float* ssim_sum, float* ws, int B, int C, int H, int W,
                              hipStream_t stream) {
  FOCR_CHECK_ARG(img1 && img2 && window_host && sq_sum && ssim_sum && ws, "null pointer");
  FOCR_CHECK_ARG(B > 0 && C >= 3 && H > 0 && W > 0, "need NCHW images with at least 3 channels");
  FOCR_CHECK_ARG(window_size > 0 && window_size <= EM_WMAX && (window_size & 1), "odd window size <= 15");
  EvalWin win;
  win.size = window_size;
  for (int i = 0; i < EM_WMAX; ++i) win.g[i] = i < window_size ? window_host[i] : 0.f;
  const int nb = cdiv(3L * H * W, 256);
  hipLaunchKernelGGL(psnr_ssim_kernel, dim3(nb, B), 256, 0, stream, img1, img2, ws, C, H, W, win);
  hipLaunchKernelGGL(psnr_ssim_fold_kernel, dim3(B), 64, 0, stream, (const float*)ws, sq_sum, ssim_sum, nb);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Input pipeline, device half of resizeNormalize (reference dataset/dataset.py:136-152): uint8 [B,H,W,3] ->
// float32 [B,3(+1),H,W]: ToTensor (x / 255, IEEE division like torch's .div(255)) and, with mask = 1, the fourth
// channel 1.0 where the pixel's luma <= the image's mean luma, else 0.0 -- PIL's convert('L') integer luma
// (19595 R + 38470 G + 7471 B + 0x8000) >> 16, mean over the image compared exactly (luma * H*W > sum of lumas).
// One block per image.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void u8_to_input_kernel(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                          int H, int W, int mask) {
  __shared__ unsigned int red[256];
  const int b = blockIdx.x, hw = H * W, C = mask ? 4 : 3;
  const unsigned char* src = in + (size_t)b * hw * 3;
  float* dst = out + (size_t)b * C * hw;
  unsigned int lsum = 0;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const unsigned int r = src[3 * p], g = src[3 * p + 1], bl = src[3 * p + 2];
    dst[p] = (float)r / 255.f;
    dst[hw + p] = (float)g / 255.f;
    dst[2 * hw + p] = (float)bl / 255.f;
    lsum += (r * 19595u + g * 38470u + bl * 7471u + 0x8000u) >> 16;
  }
  if (!mask) return;
  red[threadIdx.x] = lsum;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const unsigned long long total = red[0];
  for (int p = threadIdx.x; p < hw; p += 256) {
    const unsigned int r = src[3 * p], g = src[3 * p + 1], bl = src[3 * p + 2];
    const unsigned long long l = (r * 19595u + g * 38470u + bl * 7471u + 0x8000u) >> 16;
    dst[3 * hw + p] = (l * (unsigned long long)hw > total) ? 0.f : 1.f;      // 0 if x > mean else 255 (-> /255)
  }
}

extern "C" int focr_u8_to_input(const unsigned char* u8_nhwc, float* out_nchw, int B, int H, int W, int mask,
                                hipStream_t stream) {
  FOCR_CHECK_ARG(u8_nhwc && out_nchw && B > 0 && H > 0 && W > 0, "bad argument");
  FOCR_CHECK_ARG((long)H * W < (1l << 23), "image too large for the 32-bit luma sum");
  hipLaunchKernelGGL(u8_to_input_kernel, dim3(B), 256, 0, stream, u8_nhwc, out_nchw, H, W, mask);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

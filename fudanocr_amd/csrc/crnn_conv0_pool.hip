// First layer of the recognizer fused with its activation and pooling: Conv2d(1, 64, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2)
// (model/crnn/crnn.py:51-52: convRelu(0), pooling0), forward and data gradient, frozen-recognizer form (the training step
// needs d loss / d input only, interfaces/super_resolution.py:168-171).
//
// Unfused, the layer writes its full-resolution output ([B, 32, 128, 64] fp32 = 134 MB at B = 128), the pooling layer reads
// it back, and the backward materialises the full-resolution gradient again (pool backward -> data-gradient convolution with
// K = 576 and ONE output channel: 100 us on the implicit-GEMM kernel).  With one input channel the convolution is 9 fma per
// output: the whole chain is computed from the 2 MB input with plain fp32 VALU arithmetic (exact products, fixed summation
// order), the 134 MB tensors never exist:
//   forward : y[n, py, px, co] = max over the 2 x 2 window of relu(conv + bias), idx = window-local argmax (first strict
//             maximum in (a, b) scan order, exactly maxpool_fwd_kernel's rule);
//   backward: dx[n, iy, ix] = sum over co and the 3 x 3 taps of g[n, oy / 2, ox / 2, co] * w[co][kh][kw] for the conv outputs
//             (oy, ox) = (iy - kh + 1, ix - kw + 1) that were their window's argmax, g = dy where the pooled output is > 0
//             (the relu's backward: a window whose maximum is 0 passes nothing) -- a GATHER, no atomics, deterministic.
// HBM: forward reads 2 MB, writes 33.5 MB + 8.4 MB of idx; backward reads 33.5 + 33.5 + 8.4 MB (x 2 through L2 for the halo
// rows), writes 2 MB.
#include "focr_common.h"

#define C0_CO 64
#define C0_MAXW 128
#define C0_GP 265                 // backward: floats per channel of the staged tile (4 pooled rows x 66) + 1: conflict-free transposition

// block = (image, pair of pooled rows); thread = (channel quad, pixel lane); six input rows staged with their zero halo
__global__ __launch_bounds__(256) void crnn_conv0_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  uint8_t* __restrict__ idx, int H, int W) {
  __shared__ float rows[6][C0_MAXW + 2];
  const int OH = H >> 1, OW = W >> 1;
  const int nblk = (OH + 1) >> 1;
  const int n = blockIdx.x / nblk, P0 = (blockIdx.x - n * nblk) * 2;
  const int tid = threadIdx.x, cq = tid & 15, pl = tid >> 4;
  for (int i = tid; i < 6 * (W + 2); i += 256) {
    const int s = i / (W + 2), t = i - s * (W + 2);
    const int iy = 2 * P0 - 1 + s, ix = t - 1;
    rows[s][t] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? x[((size_t)n * H + iy) * W + ix] : 0.f;
  }
  float wr[4][9], br[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    br[e] = bias ? bias[4 * cq + e] : 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[e][k] = w[(4 * cq + e) * 9 + k];
  }
  __syncthreads();
  for (int p = pl; p < 2 * OW; p += 16) {
    const int prow = p >= OW ? 1 : 0, pcol = p - prow * OW;
    if (P0 + prow >= OH) break;
    float patch[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) patch[r][c] = rows[2 * prow + r][2 * pcol + c];
    float best[4];
    uint32_t bi = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float bv = -INFINITY;
      uint32_t bidx = 0u;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          float v = br[e];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v = fmaf(patch[a + kh][b + kw], wr[e][kh * 3 + kw], v);
          v = fmaxf(v, 0.f);
          if (v > bv) { bv = v; bidx = 2 * a + b; }
        }
      best[e] = bv;
      bi |= bidx << (8 * e);
    }
    const size_t o = (((size_t)n * OH + P0 + prow) * OW + pcol) * C0_CO + 4 * cq;
    *reinterpret_cast<float4*>(y + o) = make_float4(best[0], best[1], best[2], best[3]);
    *reinterpret_cast<uint32_t*>(idx + o) = bi;
  }
}

// block = (image, four input rows); 8 waves: wave >> 1 = (row parity, column parity) of its pixels -- the tap -> (pooled
// neighbour, window position, weight) table is then the same for the whole wave and resolved at compile time.
// Staged: the masked pooled gradient G[co][4 pooled rows][OW + 2] and the argmax bytes, transposed to channel-major so that
// the lanes of a wave (consecutive pixels) read consecutive words.
template <int PY, int PX>
__device__ __forceinline__ float c0_gather(const float* __restrict__ G, const uint8_t* __restrict__ I,
                                           const float* __restrict__ w, int base) {
  // input pixel (iy, ix) = (2 k + PY, 2 m + PX); `base` addresses pooled neighbour (k - 1 + PY, m - 1 + PX) =: (r0, c0):
  //   PY = 0: conv rows 2k-1 (r0, a = 1, kh = 2), 2k (r0 + 1, a = 0, kh = 1), 2k+1 (r0 + 1, a = 1, kh = 0)
  //   PY = 1: conv rows 2k (r0, a = 0, kh = 2), 2k+1 (r0, a = 1, kh = 1), 2k+2 (r0 + 1, a = 0, kh = 0)
  constexpr int JR[3] = {0, PY ? 0 : 1, 1};
  constexpr int AR[3] = {PY ? 0 : 1, PY ? 1 : 0, PY ? 0 : 1};
  constexpr int KH[3] = {2, 1, 0};
  constexpr int JC[3] = {0, PX ? 0 : 1, 1};
  constexpr int BC[3] = {PX ? 0 : 1, PX ? 1 : 0, PX ? 0 : 1};
  float acc = 0.f;
#pragma unroll 4
  for (int co = 0; co < C0_CO; ++co) {
    const float* g = G + co * C0_GP + base;
    const uint8_t* ib = I + co * C0_GP + base;
    float gv[2][2];
    int iv[2][2];
#pragma unroll
    for (int jr = 0; jr < 2; ++jr)
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        gv[jr][jc] = g[jr * 66 + jc];
        iv[jr][jc] = ib[jr * 66 + jc];
      }
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const float gs = iv[JR[u]][JC[v]] == 2 * AR[u] + BC[v] ? gv[JR[u]][JC[v]] : 0.f;
        acc = fmaf(gs, w[co * 9 + KH[u] * 3 + KH[v]], acc);
      }
  }
  return acc;
}

__global__ __launch_bounds__(512) void crnn_conv0_pool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ ypool, const float* __restrict__ w,
                                                                  float* __restrict__ dx, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c0_smem[];
  float* G = reinterpret_cast<float*>(c0_smem);                         // [64][C0_GP]
  uint8_t* I = c0_smem + C0_CO * C0_GP * sizeof(float);                 // [64][C0_GP]
  const int OH = H >> 1, OW = W >> 1;
  const int nblk = H >> 2;
  const int n = blockIdx.x / nblk, r = blockIdx.x - n * nblk;           // input rows 4 r .. 4 r + 3
  const int tid = threadIdx.x;
  // staged pooled row s = 0..3 <-> py = 2 r - 1 + s; staged column t = 0..OW + 1 <-> px = t - 1
  for (int i = tid; i < 4 * (OW + 2) * 16; i += 512) {
    const int cq = i & 15, pos = i >> 4, s = pos / (OW + 2), t = pos - s * (OW + 2);
    const int py = 2 * r - 1 + s, px = t - 1;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t ib = 0xffffffffu;
    if ((unsigned)py < (unsigned)OH && (unsigned)px < (unsigned)OW) {
      const size_t o = (((size_t)n * OH + py) * OW + px) * C0_CO + 4 * cq;
      const float4 d = *reinterpret_cast<const float4*>(dy + o), yp = *reinterpret_cast<const float4*>(ypool + o);
      g = make_float4(yp.x > 0.f ? d.x : 0.f, yp.y > 0.f ? d.y : 0.f, yp.z > 0.f ? d.z : 0.f, yp.w > 0.f ? d.w : 0.f);
      ib = *reinterpret_cast<const uint32_t*>(idx + o);
    }
    const int o2 = s * 66 + t;
    G[(4 * cq + 0) * C0_GP + o2] = g.x;
    G[(4 * cq + 1) * C0_GP + o2] = g.y;
    G[(4 * cq + 2) * C0_GP + o2] = g.z;
    G[(4 * cq + 3) * C0_GP + o2] = g.w;
    I[(4 * cq + 0) * C0_GP + o2] = (uint8_t)(ib & 0xffu);
    I[(4 * cq + 1) * C0_GP + o2] = (uint8_t)((ib >> 8) & 0xffu);
    I[(4 * cq + 2) * C0_GP + o2] = (uint8_t)((ib >> 16) & 0xffu);
    I[(4 * cq + 3) * C0_GP + o2] = (uint8_t)(ib >> 24);
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int combo = wave >> 1, q = (wave & 1) * 64 + lane;              // q enumerates the 2 x OW pixels of this parity class
  if (q >= 2 * OW) return;
  const int kk = q >= OW ? 1 : 0, m = q - kk * OW;                      // iy = 4 r + 2 kk + PY, ix = 2 m + PX
  const int PYv = combo >> 1, PXv = combo & 1;
  // pooled neighbour (k - 1 + PY, m - 1 + PX) with k = 2 r + kk: staged row k - 1 + PY - (2 r - 1) = kk + PY, column m + PX
  const int base = (kk + PYv) * 66 + m + PXv;
  float acc;
  if (combo == 0) acc = c0_gather<0, 0>(G, I, w, base);
  else if (combo == 1) acc = c0_gather<0, 1>(G, I, w, base);
  else if (combo == 2) acc = c0_gather<1, 0>(G, I, w, base);
  else acc = c0_gather<1, 1>(G, I, w, base);
  dx[((size_t)n * H + 4 * r + 2 * kk + PYv) * W + 2 * m + PXv] = acc;
}

static bool c0_ok(int N, int H, int W) { return N > 0 && H >= 4 && H % 4 == 0 && W >= 2 && W % 2 == 0 && W <= C0_MAXW; }

extern "C" int focr_crnn_conv0_pool_supported(int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
  return c0_ok(1, H, W) && Cin == 1 && Cout == C0_CO && KH == 3 && KW == 3 && pad == 1;
}

// x [N, H, W, 1], w [64][3][3][1] (= nn.Conv2d's [64, 1, 3, 3]), bias [64] or null -> y [N, H/2, W/2, 64], idx (uint8, same shape)
extern "C" int focr_crnn_conv0_pool_fwd(const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int N,
                                        int H, int W, hipStream_t stream) {
  FOCR_CHECK_ARG(x && w && y && idx, "null pointer");
  FOCR_CHECK_ARG(c0_ok(N, H, W), "needs H % 4 == 0, W even, W <= 128");
  hipLaunchKernelGGL(crnn_conv0_pool_fwd_kernel, dim3(N * (((H >> 1) + 1) >> 1)), 256, 0, stream, x, w, bias, y, idx, H, W);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// dy, ypool [N, H/2, W/2, 64], idx as written by the forward -> dx [N, H, W, 1] (every element written)
extern "C" int focr_crnn_conv0_pool_bwd(const float* dy, const uint8_t* idx, const float* ypool, const float* w, float* dx,
                                        int N, int H, int W, hipStream_t stream) {
  FOCR_CHECK_ARG(dy && idx && ypool && w && dx, "null pointer");
  FOCR_CHECK_ARG(c0_ok(N, H, W), "needs H % 4 == 0, W even, W <= 128");
  static const int lds = C0_CO * C0_GP * (int)(sizeof(float) + 1);
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(crnn_conv0_pool_bwd_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
      focr_set_error("focr_crnn_conv0_pool_bwd: cannot reserve %d bytes of LDS", lds);
      return FOCR_EHIP;
    }
    focr_dev_mark(attr_set);
  }
  hipLaunchKernelGGL(crnn_conv0_pool_bwd_kernel, dim3(N * (H >> 2)), 512, lds, stream, dy, idx, ypool, w, dx, H, W);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// Implicit-GEMM NHWC convolution / linear layer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Replaces the reference's torch ops nn.Conv2d / nn.Linear (SURVEY.md section 2.3 K1-K3,
// K6, K9-K12, K15; call sites model/tsrn.py:26-43,80-94,101-114, model/tbsrn.py:103-129,
// 153-163, model/stn_head.py:13-49, model/crnn/crnn.py:31-63) with ONE kernel family:
//
//   forward : Y[p][co] = act(alpha * sum_k A[p][k] * W[co][k] + bias[co] + R[p][co])
//             p = (n,oy,ox)  k = (kh,kw,ci)  A[p][k] = X[n,oy+kh-ph,ox+kw-pw,ci] (0 outside)
//   dgrad   : the same kernel on flipped/transposed weights (focr_weight_flip_transpose)
//   wgrad   : dW[co][k] = sum_p dY[p][co] * A[p][k]   (split over p, fp32 atomics)
//
// No im2col buffer exists: the A operand is gathered straight from the NHWC tensor into
// an LDS tile (rows = pixels, 32 consecutive k per chunk), the re-reads across the KHxKW
// taps are served by L2.  Precision: exact fp32 (the f32-input MFMA is a k-ordered fmaf
// chain), which is what the 1e-3 parity gate against the fp32 reference needs; roofline =
// 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Tile: 128 pixels x (32*NT) channels per 256-thread block, 4 waves stacked along pixels,
// each wave 32 pixels x 32*NT channels = NT accumulators of 16 VGPRs.  LDS pitch 36 floats:
// the ds_read_b128 fragment reads (16-lane groups, 64 banks) are conflict-free.
#include "focr_common.h"

#define BM 128
#define BK 32
#define LDP 36   // LDS row pitch in floats (BK + 4)

struct ConvGeom {
  int N, H, W, Cin;       // input  NHWC
  int OH, OW, Cout;       // output NHWC
  int KH, KW, padH, padW;
  int Ktot;               // KH*KW*Cin
  int M;                  // N*OH*OW
  int ldy;                // output row pitch (floats) >= Cout
  int ldr;                // residual row pitch
  int ldx;                // input pixel pitch (floats) >= Cin
};

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int NT, bool VEC>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const float* __restrict__ X,
                                                       const float* __restrict__ Wt,
                                                       const float* __restrict__ bias,
                                                       const float* __restrict__ R,
                                                       float* __restrict__ Y, ConvGeom g,
                                                       float alpha, int relu) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) float As[BM * LDP];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDP];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nchunks = (g.Ktot + BK - 1) / BK;

  // ---- per-thread staging coordinates -------------------------------------------------
  // VEC: thread owns float4 column c4 = tid&7 of rows (tid>>3) + 32*i, i<4
  // SCALAR: thread owns column kk = tid&31 of rows (tid>>5) + 8*i, i<16
  constexpr int AROWS = VEC ? 4 : 16;
  int rbase[AROWS];     // element offset of pixel (n, oy-ph, ox-pw, 0), only used if valid
  int riy[AROWS], rix[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int row = VEC ? (tid >> 3) + 32 * i : (tid >> 5) + 8 * i;
    int p = m0 + row;
    if (p < g.M) {
      int n = p / (g.OH * g.OW);
      int rem = p - n * (g.OH * g.OW);
      int oy = rem / g.OW, ox = rem - oy * g.OW;
      riy[i] = oy - g.padH;
      rix[i] = ox - g.padW;
      rbase[i] = n * g.H * g.W;
    } else {
      riy[i] = -100000;   // never in bounds
      rix[i] = 0;
      rbase[i] = 0;
    }
  }

  float4 areg[VEC ? 4 : 1];
  float asc[VEC ? 1 : 16];
  constexpr int BV = VEC ? (BN * BK / 4) / 256 : (BN * BK) / 256;   // per-thread B loads
  float4 breg[VEC ? BV : 1];
  float bsc[VEC ? 1 : BV];

  auto load_chunk = [&](int c) {
    const int k0 = c * BK;
    if constexpr (VEC) {
      // chunk lies inside one tap (Cin % BK == 0)
      int tap = k0 / g.Cin;
      int ci0 = k0 - tap * g.Cin + (tid & 7) * 4;
      int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int iy = riy[i] + kh, ix = rix[i] + kw;
        bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(X + ((size_t)(rbase[i] + iy * g.W + ix) * g.ldx + ci0));
        areg[i] = v;
      }
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        int idx = tid + 256 * i;
        int row = idx >> 3, c4 = idx & 7;
        int co = n0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < g.Cout) v = *reinterpret_cast<const float4*>(Wt + (size_t)co * g.Ktot + k0 + c4 * 4);
        breg[i] = v;
      }
    } else {
      int k = k0 + (tid & 31);
      bool kok = k < g.Ktot;
      int tap = k / g.Cin;
      int ci = k - tap * g.Cin;
      int kh = tap / g.KW, kw = tap - kh * g.KW;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int iy = riy[i] + kh, ix = rix[i] + kw;
        bool ok = kok && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
        asc[i] = ok ? X[(size_t)(rbase[i] + iy * g.W + ix) * g.ldx + ci] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        int row = (tid >> 5) + 8 * i;
        int co = n0 + row;
        bsc[i] = (kok && co < g.Cout) ? Wt[(size_t)co * g.Ktot + k] : 0.f;
      }
    }
  };
  auto store_chunk = [&]() {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(&As[((tid >> 3) + 32 * i) * LDP + (tid & 7) * 4]) = areg[i];
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        int idx = tid + 256 * i;
        *reinterpret_cast<float4*>(&Bs[(idx >> 3) * LDP + (idx & 7) * 4]) = breg[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) As[((tid >> 5) + 8 * i) * LDP + (tid & 31)] = asc[i];
#pragma unroll
      for (int i = 0; i < BV; ++i) Bs[((tid >> 5) + 8 * i) * LDP + (tid & 31)] = bsc[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const float* aptr = &As[(wave * 32 + li) * LDP + 4 * lh];
  const float* bptr = &Bs[li * LDP + 4 * lh];

  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    store_chunk();
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
    float4 af[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) af[t] = *reinterpret_cast<const float4*>(aptr + 8 * t);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float4 bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bf[t] = *reinterpret_cast<const float4*>(bptr + nt * 32 * LDP + 8 * t);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].x, bf[t].x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].y, bf[t].y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].z, bf[t].z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[t].w, bf[t].w, acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds channel n0 + nt*32 + li for 16 pixel rows --------------------
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int co = n0 + nt * 32 + li;
    if (co >= g.Cout) continue;
    float b = bias ? bias[co] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int p = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (p < g.M) {
        float v = alpha * acc[nt][r] + b;
        if (R) v += R[(size_t)p * g.ldr + co];
        if (relu) v = fmaxf(v, 0.f);
        Y[(size_t)p * g.ldy + co] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// wgrad:  dW[co][k] += sum_{p in split} dY[p][co] * A[p][k]
// block tile 64 co x 64 k, 4 waves as 2x2 of 32x32; reduction chunk = 32 pixels.
// ---------------------------------------------------------------------------------------
#define WP 68   // LDS pitch for the [32 pixel][64] tiles
template <bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const float* __restrict__ X,
                                                         const float* __restrict__ dY,
                                                         float* __restrict__ dW,
                                                         float* __restrict__ dbias, ConvGeom g,
                                                         int ldd, int pix_per_split) {
  __shared__ __attribute__((aligned(16))) float Ds[32 * WP];   // dY tile  [pixel][co]
  __shared__ __attribute__((aligned(16))) float Xs[32 * WP];   // A tile   [pixel][k]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = blockIdx.x * 64;
  const int co0 = blockIdx.y * 64;
  const int pbeg = blockIdx.z * pix_per_split;
  const int pend = min(g.M, pbeg + pix_per_split);
  if (pbeg >= pend) return;

  // staging: 32 pixels x 64 columns = 2048 floats = 512 float4 -> 2 per thread (VEC)
  // VEC : thread -> pixel rows (tid>>4) + 16*i, float4 column (tid&15)
  // SCALAR: thread -> column tid&63, pixel rows (tid>>6) + 4*i, i<8
  int tapkh = 0, tapkw = 0, ci0 = 0;
  bool kok = true;
  if constexpr (VEC) {
    int tap = k0 / g.Cin;                 // 64-wide k tile inside one tap (Cin % 64 == 0)
    ci0 = k0 - tap * g.Cin + (tid & 15) * 4;
    tapkh = tap / g.KW;
    tapkw = tap - tapkh * g.KW;
  } else {
    int k = k0 + (tid & 63);
    kok = k < g.Ktot;
    int tap = k / g.Cin;
    ci0 = k - tap * g.Cin;
    tapkh = tap / g.KW;
    tapkw = tap - tapkh * g.KW;
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // bias gradient (column sums of dY) rides along in the k-tile-0 blocks:
  // thread -> column tid&63, rows 8*(tid>>6) .. +7 of every staged dY tile
  const bool do_bias = dbias != nullptr && blockIdx.x == 0;
  float bsum = 0.f;
  const int wi = wave >> 1, wj = wave & 1;     // wave tile: co rows wi*32.., k cols wj*32..
  const int li = lane & 31, lh = lane >> 5;

  for (int pc = pbeg; pc < pend; pc += 32) {
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = (tid >> 4) + 16 * i;
        int p = pc + row;
        float4 dv = make_float4(0.f, 0.f, 0.f, 0.f), xv = dv;
        if (p < pend) {
          int co = co0 + (tid & 15) * 4;
          if (co + 3 < g.Cout) {
            dv = *reinterpret_cast<const float4*>(dY + (size_t)p * ldd + co);
          } else {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e = 0; e < 4; ++e)
              if (co + e < g.Cout) t[e] = dY[(size_t)p * ldd + co + e];
            dv = make_float4(t[0], t[1], t[2], t[3]);
          }
          int n = p / (g.OH * g.OW);
          int rem = p - n * (g.OH * g.OW);
          int oy = rem / g.OW, ox = rem - oy * g.OW;
          int iy = oy - g.padH + tapkh, ix = ox - g.padW + tapkw;
          if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
            xv = *reinterpret_cast<const float4*>(X + ((size_t)((n * g.H + iy) * g.W + ix) * g.ldx + ci0));
        }
        *reinterpret_cast<float4*>(&Ds[row * WP + (tid & 15) * 4]) = dv;
        *reinterpret_cast<float4*>(&Xs[row * WP + (tid & 15) * 4]) = xv;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int row = (tid >> 6) + 4 * i;
        int p = pc + row;
        float dv = 0.f, xv = 0.f;
        if (p < pend) {
          int co = co0 + (tid & 63);
          if (co < g.Cout) dv = dY[(size_t)p * ldd + co];
          if (kok) {
            int n = p / (g.OH * g.OW);
            int rem = p - n * (g.OH * g.OW);
            int oy = rem / g.OW, ox = rem - oy * g.OW;
            int iy = oy - g.padH + tapkh, ix = ox - g.padW + tapkw;
            if ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
              xv = X[(size_t)((n * g.H + iy) * g.W + ix) * g.ldx + ci0];
          }
        }
        Ds[row * WP + (tid & 63)] = dv;
        Xs[row * WP + (tid & 63)] = xv;
      }
    }
    __syncthreads();
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < 8; ++r) bsum += Ds[((tid >> 6) * 8 + r) * WP + (tid & 63)];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a = Ds[(2 * s + lh) * WP + wi * 32 + li];
      float b = Xs[(2 * s + lh) * WP + wj * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (do_bias) {   // block-uniform: fold the 4 row-group partials in LDS, then 64 atomics per block
    Ds[tid] = bsum;
    __syncthreads();
    if (tid < 64 && co0 + tid < g.Cout) atomicAdd(&dbias[co0 + tid], Ds[tid] + Ds[64 + tid] + Ds[128 + tid] + Ds[192 + tid]);
  }
  // acc[r]: row (co) = (r&3)+8*(r>>2)+4*lh, col (k) = li
  int k = k0 + wj * 32 + li;
  if (k < g.Ktot) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#ifdef WGRAD_NOATOMIC
      if (co < g.Cout) dW[(size_t)co * g.Ktot + k] = acc[r];     // timing experiment only (wrong sums)
#else
      if (co < g.Cout) atomicAdd(&dW[(size_t)co * g.Ktot + k], acc[r]);
#endif
    }
  }
}

// W[co][kh][kw][ci] -> Wd[ci][KH-1-kh][KW-1-kw][co]   (weights of the dgrad convolution)
__global__ void weight_flip_transpose_kernel(const float* __restrict__ W, float* __restrict__ Wd,
                                             int Cout, int KH, int KW, int Cin) {
  int total = Cout * KH * KW * Cin;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    // i indexes Wd: [ci][kh'][kw'][co]
    int co = i % Cout;
    int t = i / Cout;
    int kw = t % KW;
    t /= KW;
    int kh = t % KH;
    int ci = t / KH;
    Wd[i] = W[((size_t)(co * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)) * Cin + ci];
  }
}

// column sums of a [rows][C] matrix (row pitch ld): out[c] = sum_r x[r][c]  (bias grads)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     long rows, int C, int ld) {
  // block handles 64 columns x a slab of rows; threads: 64 columns x 4 row lanes
  __shared__ float red[4][64];
  int c = blockIdx.x * 64 + (threadIdx.x & 63);
  int rl = threadIdx.x >> 6;
  long rows_per = (rows + gridDim.y - 1) / gridDim.y;
  long r0 = (long)blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s = 0.f;
  if (c < C)
    for (long r = r0 + rl; r < r1; r += 4) s += x[r * ld + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) atomicAdd(&out[c], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
static int fill_geom(ConvGeom& g, int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH,
                     int padW) {
  g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.KH = KH; g.KW = KW;
  g.padH = padH; g.padW = padW;
  g.OH = H + 2 * padH - KH + 1;
  g.OW = W + 2 * padW - KW + 1;
  g.Ktot = KH * KW * Cin;
  long M = (long)N * g.OH * g.OW;
  if (g.OH <= 0 || g.OW <= 0 || M <= 0 || M > 0x7fffffffL) return -1;
  if ((long)N * H * W * Cin > 0x7fffffffL * 4L) return -1;
  g.M = (int)M;
  g.ldy = Cout; g.ldr = Cout; g.ldx = Cin;
  return 0;
}

int focr_conv_fwd_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y,
                      int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int padH, int padW,
                      int M, int ldy, int ldr, int ldx, float alpha, int relu, float* ws, long ws_floats,
                      hipStream_t stream);
long focr_conv_fwd_bx3_ws_floats(int M, int Cin, int Cout, int Ktot);

int focr_conv_wgrad_bx3(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                        int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int padH, int padW, int M,
                        int ldd, int ldx, int splits, int pps, hipStream_t stream);
long focr_conv_wgrad_bx3_ws_floats(int M, int Cin, int Cout, int Ktot);
int focr_conv3x3_c64_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                           int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx,
                           hipStream_t stream);
long focr_conv3x3_c64_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW);
// conv3x3_cin_small_wgrad.hip
int focr_conv3x3_cin_small_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                                 int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx,
                                 hipStream_t stream);
long focr_conv3x3_cin_small_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW);
// linear_wgrad.hip
int focr_linear_wgrad_eligible(long M, int K, int Cout, int ldx, int ldd);
long focr_linear_wgrad_ws_floats(long M, int K, int Cout);
int focr_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, long M, int K,
                      int Cout, int ldx, int ldd, int accumulate, hipStream_t stream);

int focr_conv9x9_cin3_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                          int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldx, int ldy,
                          float alpha, int relu, hipStream_t stream);
int focr_linear_stream_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y, int M,
                           int Cin, int Cout, int ldx, int ldy, int ldr, float alpha, int relu, uint32_t drop_k,
                           float drop_scale, uint32_t drop_seed, float mask_scale, hipStream_t stream);
extern "C" int focr_dropout(const float* x, float* y, long n, float p, uint64_t seed, hipStream_t stream);

// Tiny 1x1 layers whose contraction is not a multiple of 32 (the STN head's fc2 data gradient: [128 x 40] . [40 x 512], on
// the critical path at the end of the backward): the tiled fp32-MFMA kernel above spent 60 us on eight blocks with
// scalar loads and a barrier per 32-deep chunk.  Plain VALU: a thread owns 4 rows x 1 output column, float4 along k,
// weights and rows come from L1 / L2 (the whole problem is a few hundred KB).  fp32 fma chain in k order.
__global__ __launch_bounds__(256) void tiny_linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ res,
                                                          float* __restrict__ y, int M, int K, int Cout, int ldx, int ldy,
                                                          int ldr, float alpha, int relu) {
  const int co = blockIdx.x * 256 + threadIdx.x, m0 = blockIdx.y * 4;
  if (co >= Cout) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* wr = w + (size_t)co * K;
  const float* xr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xr[r] = x + (size_t)min(m0 + r, M - 1) * ldx;
  for (int k = 0; k < K; k += 4) {
    const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float4 xv = *reinterpret_cast<const float4*>(xr[r] + k);
      acc[r] = fmaf(xv.x, wv.x, acc[r]);
      acc[r] = fmaf(xv.y, wv.y, acc[r]);
      acc[r] = fmaf(xv.z, wv.z, acc[r]);
      acc[r] = fmaf(xv.w, wv.w, acc[r]);
    }
  }
  const float b = bias ? bias[co] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (m0 + r >= M) break;
    float v = alpha * acc[r] + b;
    if (res) v += res[(size_t)(m0 + r) * ldr + co];
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)(m0 + r) * ldy + co] = v;
  }
}

int focr_gemm_big_bx3(const float* x, const float* w, const float* bias, const float* residual, float* y, long M, int K,
                      int N, int ldx, int ldy, int ldr, float alpha, int relu, hipStream_t stream);

static int conv2d_fwd_impl(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                           int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, float alpha, int relu,
                           int ldy, int ldr, int ldx, float* ws, long ws_floats, hipStream_t stream) {
  ConvGeom g;
  FOCR_CHECK_ARG(x && w && y, "null pointer");
  FOCR_CHECK_ARG(fill_geom(g, N, H, W, Cin, Cout, KH, KW, padH, padW) == 0, "bad geometry");
  if (ldy > 0) g.ldy = ldy;
  if (ldr > 0) g.ldr = ldr;
  if (ldx > 0) g.ldx = ldx;
  FOCR_CHECK_ARG(g.ldy >= Cout && g.ldr >= Cout && g.ldx >= Cin, "row pitch too small");
  bool vec = (Cin % BK == 0) && (g.ldx % 4 == 0);
  if (focr_get_precision() != 0 &&
      focr_conv9x9_cin3_fwd(x, w, bias, residual, y, N, H, W, Cin, Cout, KH, KW, padH, padW, g.ldx, g.ldy, alpha, relu,
                            stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (vec && focr_get_precision() != 0 && KH == 1 && KW == 1 && padH == 0 && padW == 0 &&
      focr_gemm_big_bx3(x, w, bias, residual, y, g.M, Cin, Cout, g.ldx, g.ldy, g.ldr, alpha, relu, stream)) {   // gemm_big.hip
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (vec && focr_get_precision() != 0 && KH == 1 && KW == 1 && padH == 0 && padW == 0 &&
      focr_linear_stream_bx3(x, w, bias, residual, y, g.M, Cin, Cout, g.ldx, g.ldy, g.ldr, alpha, relu, 0u, 1.f, 0u,
                             0.f, stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (vec && focr_get_precision() != 0) {
    focr_conv_fwd_bx3(x, w, bias, residual, y, N, H, W, Cin, g.OH, g.OW, Cout, KH, KW, padH, padW, g.M, g.ldy,
                      g.ldr, g.ldx, alpha, relu, ws, ws_floats, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (!vec && KH == 1 && KW == 1 && padH == 0 && padW == 0 && Cin % 4 == 0 && g.ldx % 4 == 0 && Cin <= 256 &&
      (long)g.M * Cout <= (1l << 20) && cdiv(g.M, 4) <= 65535 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
    hipLaunchKernelGGL(tiny_linear_kernel, dim3(cdiv(Cout, 256), cdiv(g.M, 4)), 256, 0, stream, x, w, bias, residual, y, g.M,
                       Cin, Cout, g.ldx, g.ldy, g.ldr, alpha, relu);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  bool wide = Cout > 32;
  dim3 grid(cdiv(g.M, BM), cdiv(Cout, wide ? 64 : 32));
  if (vec && wide)
    hipLaunchKernelGGL((conv_fwd_kernel<2, true>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  else if (vec)
    hipLaunchKernelGGL((conv_fwd_kernel<1, true>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  else if (wide)
    hipLaunchKernelGGL((conv_fwd_kernel<2, false>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  else
    hipLaunchKernelGGL((conv_fwd_kernel<1, false>), grid, 256, 0, stream, x, w, bias, residual, y, g, alpha, relu);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_conv2d_fwd(const float* x, const float* w, const float* bias,
                               const float* residual, float* y, int N, int H, int W, int Cin,
                               int Cout, int KH, int KW, int padH, int padW, float alpha, int relu,
                               int ldy, int ldr, int ldx, hipStream_t stream) {
  return conv2d_fwd_impl(x, w, bias, residual, y, N, H, W, Cin, Cout, KH, KW, padH, padW, alpha, relu, ldy, ldr, ldx,
                         nullptr, 0, stream);
}
// Same convolution with caller-provided scratch: layers with few output tiles and a long contraction are split along K
// (deterministic slot reduction).  focr_conv2d_fwd_ws_floats returns the scratch size (0: the layer does not split).
extern "C" long focr_conv2d_fwd_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW) {
  ConvGeom g;
  if (focr_get_precision() == 0 || fill_geom(g, N, H, W, Cin, Cout, KH, KW, padH, padW) != 0 || Cin % BK != 0) return 0;
  if (KH == 9 || (KH == 1 && KW == 1 && g.M >= 16384 && (Cin == 64 || Cin == 128))) return 0;   // special kernels
  return focr_conv_fwd_bx3_ws_floats(g.M, Cin, Cout, g.Ktot);
}
extern "C" int focr_conv2d_fwd_ws(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                  int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW,
                                  float alpha, int relu, int ldy, int ldr, int ldx, float* ws, long ws_floats,
                                  hipStream_t stream) {
  return conv2d_fwd_impl(x, w, bias, residual, y, N, H, W, Cin, Cout, KH, KW, padH, padW, alpha, relu, ldy, ldr, ldx, ws,
                         ws_floats, stream);
}

// y = Dropout_p(relu(alpha * x W^T + b)) for a Linear (rows x Cin -> rows x Cout), the FFN front half
// (PositionwiseFeedForward, tbsrn.py:162-163).  P(keep) is quantised to 1/65536; *keep_scale receives 1/P(keep),
// the factor relu_bwd_scaled needs (dropped elements are exactly the zeros of y, so the backward needs no mask).
// Fused into the streaming kernel's epilogue when the layer qualifies, otherwise linear + in-place dropout.
extern "C" int focr_linear_relu_dropout_fwd(const float* x, const float* w, const float* bias, float* y, long rows,
                                            int Cin, int Cout, float alpha, float p_drop, uint64_t seed,
                                            float* keep_scale, hipStream_t stream) {
  FOCR_CHECK_ARG(x && w && y && rows > 0 && Cin > 0 && Cout > 0 && keep_scale, "bad argument");
  FOCR_CHECK_ARG(p_drop > 0.f && p_drop < 1.f, "bad dropout probability");
  const uint32_t kq = 65536u - (uint32_t)(p_drop * 65536.0f + 0.5f);
  FOCR_CHECK_ARG(kq > 0u, "dropout probability rounds to 1");
  *keep_scale = 65536.f / (float)kq;
  if (focr_get_precision() != 0 && rows < (1l << 31) &&
      focr_linear_stream_bx3(x, w, bias, nullptr, y, (int)rows, Cin, Cout, Cin, Cout, Cout, alpha, 1, kq, *keep_scale,
                             (uint32_t)(seed ^ (seed >> 32)), 0.f, stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  int rc = focr_conv2d_fwd(x, w, bias, nullptr, y, (int)rows, 1, 1, Cin, Cout, 1, 1, 0, 0, alpha, 1, 0, 0, 0, stream);
  if (rc != FOCR_OK) return rc;
  return focr_dropout(y, y, rows * Cout, 1.f - (float)kq / 65536.f, seed, stream);
}

// g = (h > 0) ? scale * (dy W^T) : 0 -- the data gradient of a Linear whose INPUT h is the output of a relu (or fused
// relu + dropout: dropped elements are the zeros of h) Linear, with that producer's relu backward fused into the epilogue
// (one pass instead of dgrad + relu_bwd_scaled: the intermediate dy_h is never written).  w: [Cout][Cin] as for
// focr_conv2d_fwd (i.e. the TRANSPOSED weight of the layer whose data gradient this is).  FOCR_EUNSUPPORTED when the
// streaming kernel does not take the shape (the caller then runs the two passes).
extern "C" int focr_linear_masked_fwd(const float* x, const float* w, const float* mask_src, float* y, long rows,
                                      int Cin, int Cout, float scale, hipStream_t stream) {
  FOCR_CHECK_ARG(x && w && mask_src && y && rows > 0 && scale != 0.f, "bad argument");
  if (focr_get_precision() != 0 && rows < (1l << 31) &&
      focr_linear_stream_bx3(x, w, nullptr, mask_src, y, (int)rows, Cin, Cout, Cin, Cout, Cout, 1.f, 0, 0u, 1.f, 0u, scale,
                             stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  focr_set_error("focr_linear_masked_fwd: shape not taken by the streaming kernel");
  return FOCR_EUNSUPPORTED;
}

// dw must hold Cout*KH*KW*Cin floats, dbias (nullable) Cout floats.  The kernel ACCUMULATES with fp32
// atomics: prezeroed = 0 -> the buffers are cleared here first (overwrite semantics); prezeroed = 1 ->
// the caller guarantees they are zero (e.g. slices of a gradient buffer zeroed once per step).
// ws / ws_floats (optional): focr_conv2d_wgrad_ws_floats() floats of scratch; with it the layers that have a
// partial-tile path (3x3, Cin = 64, bf16x3 modes) use no atomics at all.
extern "C" long focr_conv2d_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH,
                                            int padW) {
  if (focr_get_precision() == 0) return 0;
  // Which kernel takes the layer is decided again at launch time with the caller's row pitches (ldd / ldx), which this
  // query does not see: the workspace is sized for the LARGEST candidate, so that whichever path the launch falls to
  // finds room (every path also checks ws_floats against its own need before touching the buffer).
  long n = 0;
  if (KH == 1 && KW == 1 && padH == 0 && padW == 0 && focr_linear_wgrad_eligible((long)N * H * W, Cin, Cout, 4, 4))
    n = focr_linear_wgrad_ws_floats((long)N * H * W, Cin, Cout);
  long m = focr_conv3x3_c64_ws_floats(N, H, W, Cin, Cout, KH, KW, padH, padW);
  if (m > n) n = m;
  m = focr_conv3x3_cin_small_ws_floats(N, H, W, Cin, Cout, KH, KW, padH, padW);
  if (m > n) n = m;
  ConvGeom g;
  if (fill_geom(g, N, H, W, Cin, Cout, KH, KW, padH, padW) != 0) return n;
  m = focr_conv_wgrad_bx3_ws_floats(g.M, Cin, Cout, g.Ktot);
  return m > n ? m : n;
}

extern "C" int focr_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N,
                                 int H, int W, int Cin, int Cout, int KH, int KW, int padH,
                                 int padW, int ldd, int ldx, int prezeroed, float* ws, long ws_floats,
                                 hipStream_t stream) {
  ConvGeom g;
  FOCR_CHECK_ARG(x && dy && dw, "null pointer");
  FOCR_CHECK_ARG(fill_geom(g, N, H, W, Cin, Cout, KH, KW, padH, padW) == 0, "bad geometry");
  if (ldd <= 0) ldd = Cout;
  if (ldx > 0) g.ldx = ldx;
  FOCR_CHECK_ARG(ldd >= Cout && g.ldx >= Cin, "row pitch too small");
  // the transformer linears (1x1, 128-multiples, many rows): streaming kernel + fixed-order slot reduction
  // (linear_wgrad.hip), no memset and no atomics
  if (KH == 1 && KW == 1 && padH == 0 && padW == 0 && focr_get_precision() != 0 && ws &&
      focr_get_tuning(FOCR_TUNE_LINEAR_WGRAD_STREAM) && focr_linear_wgrad_eligible(g.M, Cin, Cout, g.ldx, ldd) &&
      ws_floats >= focr_linear_wgrad_ws_floats(g.M, Cin, Cout)) {
    focr_linear_wgrad(x, dy, dw, dbias, ws, ws_floats, g.M, Cin, Cout, g.ldx, ldd, prezeroed, stream);
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  // 3x3 with <= 4 input channels and 32 output channels (first STN layer): partial sums + fixed-order fold, overwrites dw / dbias
  if (focr_get_precision() != 0 &&
      focr_conv3x3_cin_small_wgrad(x, dy, dw, dbias, ws, ws_floats, N, H, W, Cin, Cout, KH, KW, padH, padW, ldd, g.ldx, stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (!prezeroed && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * g.Ktot, stream) != hipSuccess) {
    focr_set_error("focr_conv2d_wgrad: memset failed");
    return FOCR_EHIP;
  }
  bool vec = (Cin % 64 == 0) && (ldd % 4 == 0) && (g.ldx % 4 == 0);
  int tiles = cdiv(g.Ktot, 64) * cdiv(Cout, 64);
#ifndef WGRAD_BLOCKS
#define WGRAD_BLOCKS 2048
#endif
  int splits = WGRAD_BLOCKS / tiles;
  int maxsplits = cdiv(g.M, 256);              // >= 8 reduction chunks per block
  if (splits > maxsplits) splits = maxsplits;
  if (splits < 1) splits = 1;
  int pps = cdiv(cdiv(g.M, splits), 64) * 64;
  splits = cdiv(g.M, pps);
  dim3 grid(cdiv(g.Ktot, 64), cdiv(Cout, 64), splits);
  if (dbias && !prezeroed && hipMemsetAsync(dbias, 0, sizeof(float) * Cout, stream) != hipSuccess) {
    focr_set_error("focr_conv2d_wgrad: memset failed");
    return FOCR_EHIP;
  }
  if (vec && focr_get_precision() != 0 &&
      focr_conv3x3_c64_wgrad(x, dy, dw, dbias, ws, ws_floats, N, H, W, Cin, Cout, KH, KW, padH, padW, ldd, g.ldx,
                             stream)) {
    FOCR_LAUNCH_CHECK();
    return FOCR_OK;
  }
  if (vec && focr_get_precision() != 0)
    focr_conv_wgrad_bx3(x, dy, dw, dbias, ws, ws_floats, N, H, W, Cin, g.OH, g.OW, Cout, KH, KW, padH, padW, g.M, ldd,
                        g.ldx, splits, pps, stream);
  else if (vec)
    hipLaunchKernelGGL((conv_wgrad_kernel<true>), grid, 256, 0, stream, x, dy, dw, dbias, g, ldd, pps);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<false>), grid, 256, 0, stream, x, dy, dw, dbias, g, ldd, pps);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

// All layers of a model in ONE launch (the weights do not change during backward, so the training step flips every
// layer once, up front, instead of launching a ~6 us kernel in front of each data-gradient GEMM).
// descs_dev: device array of n focr_flip_desc (include/focr.h).
struct FlipDesc {
  const float* w;
  float* wd;
  int cout, kh, kw, cin;
};
__global__ __launch_bounds__(256) void weight_flip_transpose_batched_kernel(const FlipDesc* __restrict__ descs) {
  const FlipDesc d = descs[blockIdx.y];
  const int total = d.cout * d.kh * d.kw * d.cin;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int co = i % d.cout;
    int t = i / d.cout;
    int kw = t % d.kw;
    t /= d.kw;
    int kh = t % d.kh;
    int ci = t / d.kh;
    d.wd[i] = d.w[((size_t)(co * d.kh + (d.kh - 1 - kh)) * d.kw + (d.kw - 1 - kw)) * d.cin + ci];
  }
}
extern "C" int focr_weight_flip_transpose_batched(const void* descs_dev, int n, int max_elems, hipStream_t stream) {
  FOCR_CHECK_ARG(descs_dev && n > 0 && max_elems > 0, "bad argument");
  int gx = cdiv(max_elems, 256 * 8);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(weight_flip_transpose_batched_kernel, dim3(gx, n), 256, 0, stream,
                     reinterpret_cast<const FlipDesc*>(descs_dev));
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_weight_flip_transpose(const float* w, float* wd, int Cout, int KH, int KW,
                                          int Cin, hipStream_t stream) {
  FOCR_CHECK_ARG(w && wd, "null pointer");
  int total = Cout * KH * KW * Cin;
  hipLaunchKernelGGL(weight_flip_transpose_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), 256, 0,
                     stream, w, wd, Cout, KH, KW, Cin);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_colsum(const float* x, float* out, long rows, int C, int ld, hipStream_t stream) {
  FOCR_CHECK_ARG(x && out && rows > 0 && C > 0, "bad argument");
  if (ld <= 0) ld = C;
  if (hipMemsetAsync(out, 0, sizeof(float) * C, stream) != hipSuccess) {
    focr_set_error("focr_colsum: memset failed");
    return FOCR_EHIP;
  }
  int ry = cdiv(rows, 512);
  if (ry > 256) ry = 256;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 64), ry), 256, 0, stream, x, out, rows, C, ld);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

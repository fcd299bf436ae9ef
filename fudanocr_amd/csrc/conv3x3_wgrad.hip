// Weight gradient of the 3x3 / pad 1 / Cin = 64 layers (SRB conv1/conv2, block7, upsample conv; tsrn.py:89-98,110-114,
// tbsrn.py:246-257), bf16x3, streaming over image rows.
//
// The generic wgrad kernel (conv_bx3.hip) launches one block per (tap, pixel range): every dY and X element is fetched
// nine times through L2, and that traffic IS its run time (106 us with, 51 us without the global loads at B = 128).
// Here one block owns a 64-channel output slice and ALL nine taps for a range of input rows:
//   * accumulators of the nine 64x64 tap tiles stay in registers (4 waves x 9 x 32x32);
//   * per input row iy the X row is fetched once and staged TRANSPOSED (Xt[ci][px], bf16 hi/lo) three times, shifted
//     by kw - 1 pixels -- the shifted 4-pixel quads are assembled with DPP row shifts (lane = pixel quad), so every
//     LDS store and every MFMA fragment read stays 8/16-byte aligned; the zero halo falls out of bound_ctrl;
//   * tap row kh pairs input row iy with dY row iy + 1 - kh: the three dY rows live in a rolling 3-slot LDS window,
//     one new row per step;
//   * next step's X / dY rows are prefetched into registers during the 108 MFMAs of the current one.
// HBM/L2 traffic: X once, dY (1 + 2/rows_per_block) times.
#include "focr_common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8;

#define GP 72            // LDS pitch (bf16) of a transposed tile row: 64 pixels + 8
#define TILE_E (64 * GP) // elements of one [64][GP] plane

// thread (pq = pixel quad, c4 = first of 4 channels) holds a 4 px x 4 ch patch; for channel e returns the packed
// bf16 pixels (w0 = px0 | px1 << 16, w1 = px2 | px3 << 16) of the hi and lo planes
__device__ __forceinline__ void pack_quad(const float4 (&r)[4], int e, uint32_t& h0, uint32_t& h1, uint32_t& l0,
                                          uint32_t& l1) {
  const float v[4] = {e == 0 ? r[0].x : e == 1 ? r[0].y : e == 2 ? r[0].z : r[0].w,
                      e == 0 ? r[1].x : e == 1 ? r[1].y : e == 2 ? r[1].z : r[1].w,
                      e == 0 ? r[2].x : e == 1 ? r[2].y : e == 2 ? r[2].z : r[2].w,
                      e == 0 ? r[3].x : e == 1 ? r[3].y : e == 2 ? r[3].z : r[3].w};
  focr_bf16x2 ha, la, hb, lb;
  focr_split2(f32x2{v[0], v[1]}, ha, la);
  focr_split2(f32x2{v[2], v[3]}, hb, lb);
  h0 = __builtin_bit_cast(uint32_t, ha);
  h1 = __builtin_bit_cast(uint32_t, hb);
  l0 = __builtin_bit_cast(uint32_t, la);
  l1 = __builtin_bit_cast(uint32_t, lb);
}
__device__ __forceinline__ void st2(__bf16* p, uint32_t a, uint32_t b) {
  *reinterpret_cast<uint2*>(p) = make_uint2(a, b);
}
// value of the previous / next lane of the 16-lane row (0 at the row ends): the 1-pixel halo of a 64-pixel image row
__device__ __forceinline__ uint32_t from_prev(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t from_next(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101 /*row_shl:1*/, 0xf, 0xf, true);
}

// plain transposed staging of a dY row: Dt[co][px]
__device__ __forceinline__ void stage_d(__bf16* Dh, __bf16* Dl, int c4, int pq, const float4 (&r)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t h0, h1, l0, l1;
    pack_quad(r, e, h0, h1, l0, l1);
    st2(&Dh[(c4 + e) * GP + 4 * pq], h0, h1);
    st2(&Dl[(c4 + e) * GP + 4 * pq], l0, l1);
  }
}
// X row: three copies, copy kw holds X[px + kw - 1] at position px
// keep_prev / keep_next (all ones or zero): the lane's left / right neighbour belongs to the same image row.  With several
// narrow images side by side in the 64-pixel chunk (WIDE kernel below) the first / last quad lane of an image segment must
// see a ZERO halo pixel, not the neighbouring image's edge.
__device__ __forceinline__ void stage_x(__bf16* Xh, __bf16* Xl, int c4, int pq, const float4 (&r)[4],
                                        uint32_t keep_prev = 0xffffffffu, uint32_t keep_next = 0xffffffffu) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    uint32_t h0, h1, l0, l1;
    pack_quad(r, e, h0, h1, l0, l1);
    const int o = (c4 + e) * GP + 4 * pq;
    st2(&Xh[TILE_E + o], h0, h1);
    st2(&Xl[TILE_E + o], l0, l1);
    st2(&Xh[o], __builtin_amdgcn_alignbit(h0, from_prev(h1) & keep_prev, 16), __builtin_amdgcn_alignbit(h1, h0, 16));
    st2(&Xl[o], __builtin_amdgcn_alignbit(l0, from_prev(l1) & keep_prev, 16), __builtin_amdgcn_alignbit(l1, l0, 16));
    st2(&Xh[2 * TILE_E + o], __builtin_amdgcn_alignbit(h1, h0, 16), __builtin_amdgcn_alignbit(from_next(h0) & keep_next, h1, 16));
    st2(&Xl[2 * TILE_E + o], __builtin_amdgcn_alignbit(l1, l0, 16), __builtin_amdgcn_alignbit(from_next(l0) & keep_next, l1, 16));
  }
}

#ifdef C3W_TIMING      // timing-only builds: s_memtime stamps of wave 0 of every block (tools/dev/c3w_timing.py)
__device__ unsigned long long c3w_stamps[1024][16];
extern "C" int focr_debug_c3w_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(c3w_stamps), sizeof(c3w_stamps)) == hipSuccess ? 0 : -1;
}
#define C3W_STAMP(i) do { if (tid == 0 && blockIdx.y < 1024) c3w_stamps[blockIdx.y][i] = __builtin_readcyclecounter(); } while (0)
#else
#define C3W_STAMP(i) do { } while (0)
#endif

// PART != nullptr: the block writes its 9 x 64 x 64 partial tile to PART[blockIdx.y][co tile] with plain stores (no
// same-address atomics: 256 blocks adding into one 147 KB array cost as much as the rest of the kernel) and
// conv3x3_c64_reduce_kernel folds them.  PART == nullptr: fp32 atomics into dW.
// WIDE (round 6): the same kernel for Cin = 64 * gridDim.z on NARROW maps (the SLD / text-focus ResNets: 256 - 1024
// channels on 16 x 16 and 8 x 8 maps).  A row step's 64-pixel chunk holds AB = 64 / W images side by side -- row iy of
// images AB n .. AB n + AB - 1 --, so that the chunk is full (one image row alone would idle three quarters of every
// MFMA at W = 16); "rows" count image GROUPS x H; the vertical structure (dY window, tap rows) is the same for all AB images;
// the horizontal halo between neighbouring images is zeroed in stage_x.  blockIdx.z picks the 64-channel input slice;
// partial tiles always go to PART (conv3x3_wide_reduce_kernel folds them into the [Cout][9][Cin] gradient).
template <bool WIDE>
__global__ __launch_bounds__(256) void conv3x3_c64_wgrad_kernel_t(const float* __restrict__ X,
                                                                  const float* __restrict__ dY, float* __restrict__ dW,
                                                                  float* __restrict__ dbias, float* __restrict__ PART,
                                                                  int N, int H, int W, int Cout, int ldx, int ldd,
                                                                  int rows_per_block, int AB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
  __bf16* Xh = reinterpret_cast<__bf16*>(smem3);       // [3 kw copies][64 ci][GP]
  __bf16* Xl = Xh + 3 * TILE_E;
  __bf16* Dh = Xl + 3 * TILE_E;                        // [3 slots][64 co][GP]
  __bf16* Dl = Dh + 3 * TILE_E;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  C3W_STAMP(0);
  const int co0 = blockIdx.x * 64;
  const int ci0 = WIDE ? blockIdx.z * 64 : 0;
  const int R0 = blockIdx.y * rows_per_block, R1 = min((WIDE ? N / AB : N) * H, R0 + rows_per_block);
  if (R0 >= R1) return;
  const int pq = tid & 15, c4 = (tid >> 4) * 4;
  const bool pxok = WIDE ? true : 4 * pq < W;          // W % 4 == 0 (WIDE: AB * W == 64, every quad lane is a pixel quad)
  // WIDE: this lane's image inside the group and its pixel offset there; pixel index of (group row G) = rowpix(G) + apix
  const int seg = WIDE ? W / 4 : 16, aimg = WIDE ? pq / seg : 0;
  const int apix = WIDE ? aimg * H * W + (pq % seg) * 4 : 4 * pq;
  const uint32_t keep_prev = (WIDE && pq % seg == 0) ? 0u : 0xffffffffu;
  const uint32_t keep_next = (WIDE && pq % seg == seg - 1) ? 0u : 0xffffffffu;
  auto rowpix = [&](int G) -> size_t {
    return WIDE ? (size_t)((G / H) * AB * H + G % H) * W : (size_t)G * W;
  };
  const int wi = wave >> 1, wj = wave & 1;
  const int aoff = (wi * 32 + li) * GP + 8 * lh, boff = (wj * 32 + li) * GP + 8 * lh;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  float4 xr[4], dr[4];
#define LOAD_X(G)                                                                                        \
  {                                                                                                      \
    const float* p_ = X + (rowpix(G) + apix) * ldx + ci0 + c4;                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) xr[j] =                                                \
        pxok ? *reinterpret_cast<const float4*>(p_ + (size_t)j * ldx) : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
#define LOAD_D_TO(G, R_)                                                                                 \
  {                                                                                                      \
    const float* p_ = dY + (rowpix(G) + apix) * ldd + co0 + c4;                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) R_[j] =                                                \
        pxok ? *reinterpret_cast<const float4*>(p_ + (size_t)j * ldd) : make_float4(0.f, 0.f, 0.f, 0.f); \
  }
#define STAGE_D_FROM(G, R_)                                                                              \
  {                                                                                                      \
    const int s_ = (G) % 3;                                                                              \
    stage_d(Dh + s_ * TILE_E, Dl + s_ * TILE_E, c4, pq, R_);                                             \
    if (dbias && (G) >= R0 && (G) < R1) {                                                                \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                    \
        bsum[0] += R_[j].x; bsum[1] += R_[j].y; bsum[2] += R_[j].z; bsum[3] += R_[j].w;                  \
      }                                                                                                  \
    }                                                                                                    \
  }
// a dY row outside the image: zeros in its window slot, so that the product loop below needs no row conditions (straight-line
// code: hipcc hoists the fragment reads over the matrix instructions; with `if (row valid)` around every tap row each of the
// twelve (k step, tap row) groups waited out its own LDS reads)
#define ZERO_D(G)                                                                                        \
  {                                                                                                      \
    const int s_ = (G) % 3;                                                                              \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                      \
      st2(&Dh[s_ * TILE_E + (c4 + e) * GP + 4 * pq], 0u, 0u);                                            \
      st2(&Dl[s_ * TILE_E + (c4 + e) * GP + 4 * pq], 0u, 0u);                                            \
    }                                                                                                    \
  }
#define LOAD_D(G) LOAD_D_TO(G, dr)
#define STAGE_D(G) STAGE_D_FROM(G, dr)

  bool have_x = false, have_d = false;                 // registers hold X row g / dY row g + 1 of the coming step
  for (int g = R0; g < R1; ++g) {
    const int iy = g % H;
    const bool first = g == R0;
    // dY rows of this step: g + 1 - kh, valid while inside the image
    const bool v0 = iy + 1 < H, v1 = true, v2 = iy > 0;
#if defined(C3W_ABL_LOAD) || defined(C3W_ABL_STAGE)     // timing-only builds: loads / staging of the first step only
#ifdef C3W_ABL_LOAD
#define C3W_LD first
#define C3W_ST true
#else
#define C3W_LD true
#define C3W_ST first
#endif
#else
#define C3W_LD true
#define C3W_ST true
#endif
    if (first || iy == 0) {                            // window not primed (block start / new image): rows g-1, g
      // ALL requests of the priming step go out before the first of them is consumed (round 4: they were issued and
      // waited for one after the other -- four dependent HBM round trips, ~10 of the kernel's 43 us with 8 rows per block)
      const bool pa = v2 && first;
      float4 da[4], db[4];
      LOAD_D_TO(pa ? g - 1 : g, da)
      if (C3W_LD) LOAD_D_TO(g, db)
      if (v0 && !have_d && C3W_LD) { LOAD_D(g + 1) have_d = true; }
      if (!have_x && C3W_LD) { LOAD_X(g) have_x = true; }
      if (pa) STAGE_D_FROM(g - 1, da)
      if (C3W_ST) STAGE_D_FROM(g, db)
    }
    if (v0) {
      if (!have_d && C3W_LD) LOAD_D(g + 1)
      if (C3W_ST) STAGE_D(g + 1)
    } else {
      ZERO_D(g + 1)                                    // below the image: the tap row kh = 0 multiplies zeros
    }
    if (!v2) ZERO_D(g + 2)                             // above the image (slot of row g - 1): tap row kh = 2 likewise
    if (!have_x && C3W_LD) LOAD_X(g)
    if (C3W_ST) stage_x(Xh, Xl, c4, pq, xr, keep_prev, keep_next);
    if (g == R0 + 1) C3W_STAMP(4);                     // second step staged (before its barrier)
    __syncthreads();
    if (first) C3W_STAMP(1);                           // primed: first barrier passed
    if (g == R0 + 1) C3W_STAMP(5);
    have_x = g + 1 < R1;
#ifdef C3W_ABL_LOAD
    have_x = have_d = false;
#else
    if (have_x) LOAD_X(g + 1)
    have_d = g + 1 < R1 && (g + 1) % H + 1 < H && (g + 1) % H != 0;   // next step stages row g + 2 from registers
    if (have_d) LOAD_D(g + 2)
#endif
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      wbf16x8 bh[3], bl[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        bh[kw] = *reinterpret_cast<const wbf16x8*>(&Xh[kw * TILE_E + boff + 16 * m]);
        bl[kw] = *reinterpret_cast<const wbf16x8*>(&Xl[kw * TILE_E + boff + 16 * m]);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int s = (g + 4 - kh) % 3;                 // slot of dY row g + 1 - kh (row -1 of the first image: slot 2, zeroed)
        wbf16x8 ah = *reinterpret_cast<const wbf16x8*>(&Dh[s * TILE_E + aoff + 16 * m]);
        wbf16x8 al = *reinterpret_cast<const wbf16x8*>(&Dl[s * TILE_E + aoff + 16 * m]);
#ifdef C3W_ABL_MFMA
        acc[kh * 3][0] += (float)ah[0] + (float)al[1] + (float)bh[0][0] + (float)bl[1][1] + (float)bh[2][2] + (float)bl[2][3] + (float)bh[1][4] + (float)bl[0][5];
        continue;
#endif
        // product-major order: consecutive matrix instructions write different accumulators (each accumulator still
        // receives hi*hi, hi*lo, lo*hi in that order)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[kw], acc[kh * 3 + kw], 0, 0, 0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[kw], acc[kh * 3 + kw], 0, 0, 0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[kw], acc[kh * 3 + kw], 0, 0, 0);
      }
    }
    if (first) C3W_STAMP(2);                           // first step's products issued
    __syncthreads();
    if (first) C3W_STAMP(3);
  }
  C3W_STAMP(6);
  if (dbias && (!WIDE || blockIdx.z == 0)) {     // fold the 16 pixel-quad lanes of each column: 64 atomics per block
    float* red = reinterpret_cast<float*>(Xh);
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pq * 64 + c4 + e] = bsum[e];
    __syncthreads();
    if (tid < 64) {
      float s2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s2 += red[r * 64 + tid];
      atomicAdd(&dbias[co0 + tid], s2);
    }
  }
  const int ci = wj * 32 + li;
#ifdef C3W_ABL_EPI
  if (acc[0][0] != 12345.f) return;
#endif
  if (PART) {
    // slot layout = the dW slice of this co tile: [64 co][9][64 ci]
    float* slot = PART + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z + blockIdx.z) * (64 * 9 * 64);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        slot[((size_t)col * 9 + t) * 64 + ci] = acc[t][r];
      }
    C3W_STAMP(7);
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      atomicAdd(&dW[((size_t)co * 9 + t) * 64 + ci], acc[t][r]);
    }
}

// WIDE: dW[co][tap][Cin] = sum over the row blocks of PART[row block][co tile][ci tile][64 co][9][64 ci], in row-block order
__global__ __launch_bounds__(256) void conv3x3_wide_reduce_kernel(const float* __restrict__ PART, float* __restrict__ dW,
                                                                  int ntco, int ntci, int nb, int accumulate) {
  const int Cin = ntci * 64;
  const long n4 = (long)ntco * 64 * 9 * Cin / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;          // float4 index into dW
  if (i >= n4) return;
  const long e0 = i * 4;
  const int ci = (int)(e0 % Cin), t = (int)((e0 / Cin) % 9), co = (int)(e0 / ((long)9 * Cin));
  const size_t in_slot = ((size_t)(co & 63) * 9 + t) * 64 + (ci & 63);
  const size_t tile = (size_t)(co >> 6) * ntci + (ci >> 6);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = 0; b < nb; b += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = b + u < nb ? *reinterpret_cast<const float4*>(PART + (((size_t)(b + u) * ntco * ntci + tile) * (64 * 9 * 64) + in_slot))
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  float4* o = reinterpret_cast<float4*>(dW) + i;
  if (accumulate) {
    const float4 p = *o;
    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
  }
  *o = s;
}

// dW[co tile][i] = sum over the row blocks of PART[row block][co tile][i], in row-block order (round 4: was eight groups
// of blocks adding their sums with one fp32 atomic per element -- 1.2 M same-address-contended atomics per launch and a
// result that depended on their order; 49-67 us in the step for 38 MB of partials).  Block = 8 float4 elements x 32
// groups of row blocks; every thread keeps its group's loads in flight together, the groups are folded through LDS.
// dW is OVERWRITTEN (the callers hand over a zeroed buffer: same result, no dependence on it).
#define C3_RG 32
__global__ __launch_bounds__(256) void conv3x3_c64_reduce_kernel(const float* __restrict__ PART, float* __restrict__ dW,
                                                                 int ntile, int nb) {
  __shared__ float4 red[C3_RG][8];
  const long per = 64 * 9 * 64 / 4;                      // float4 per co tile
  const int e8 = threadIdx.x & 7, g = threadIdx.x >> 3;
  const long i = (long)blockIdx.x * 8 + e8;              // ntile * per is a multiple of 8
  const int tile = (int)(i / per);
  const long e = i - tile * per;
  const int pg = (nb + C3_RG - 1) / C3_RG, b0 = g * pg, b1 = min(nb, b0 + pg);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b = b0; b < b1; b += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = b + u < b1 ? reinterpret_cast<const float4*>(PART)[((size_t)(b + u) * ntile + tile) * per + e]
                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  red[g][e8] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    float4 t = red[0][e8];
#pragma unroll
    for (int q = 1; q < C3_RG; ++q) { t.x += red[q][e8].x; t.y += red[q][e8].y; t.z += red[q][e8].z; t.w += red[q][e8].w; }
    reinterpret_cast<float4*>(dW)[i] = t;
  }
}

// WIDE applicability: 3x3 / pad 1, Cin and Cout multiples of 64 (Cin > 64), a map width that packs the 64-pixel chunk with
// whole images (W in {8, 16, 32}), a batch that is a multiple of that packing, enough rows to fill the chip
static int c3_wide_ab(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx) {
  static const bool on = !(getenv("FOCR_C3W_WIDE") && getenv("FOCR_C3W_WIDE")[0] == '0');
  if (!on || !(KH == 3 && KW == 3 && padH == 1 && padW == 1 && Cin > 64 && Cin % 64 == 0 && Cout % 64 == 0 &&
               (W == 8 || W == 16 || W == 32) && ldx % 4 == 0 && ldd % 4 == 0))
    return 0;
  const int ab = 64 / W;
  if (N % ab != 0 || (long)(N / ab) * H < 16) return 0;
  return ab;
}
static void c3_wide_blocks(int N, int H, int Cin, int Cout, int ab, int& nb, int& rpb) {
  const int rows = (N / ab) * H, pairs = (Cin / 64) * (Cout / 64);
  nb = (256 + pairs - 1) / pairs;                 // >= one block per CU (110 KB of LDS each)
  if (nb < 1) nb = 1;
  if (nb > rows / 8) nb = rows / 8 > 0 ? rows / 8 : 1;      // at least 8 row steps per block (priming + epilogue amortised)
  rpb = cdiv(rows, nb);
  nb = cdiv(rows, rpb);
}
static bool c3_applicable(int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx) {
  return KH == 3 && KW == 3 && padH == 1 && padW == 1 && Cin == 64 && Cout % 64 == 0 && W <= 64 && W % 4 == 0 &&
         ldx % 4 == 0 && ldd % 4 == 0;
}
static void c3_blocks(int N, int H, int Cout, int& nb, int& rpb) {
  const int rows = N * H;
  nb = 256 / (Cout / 64);                         // one block per CU (110 KB of LDS each)
  if (nb < 1) nb = 1;
  if (nb > rows) nb = rows;
  rpb = cdiv(rows, nb);
  nb = cdiv(rows, rpb);
}

// floats of workspace that make the 3x3 / Cin 64 weight gradient atomic-free (0: layer not handled by this file)
long focr_conv3x3_c64_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW) {
  if (const int ab = c3_wide_ab(N, H, W, Cin, Cout, KH, KW, padH, padW, 4, 4)) {
    int nb, rpb;
    c3_wide_blocks(N, H, Cin, Cout, ab, nb, rpb);
    return (long)nb * (Cout / 64) * (Cin / 64) * (64 * 9 * 64);
  }
  if (!c3_applicable(W, Cin, Cout, KH, KW, padH, padW, 4, 4)) return 0;
  int nb, rpb;
  c3_blocks(N, H, Cout, nb, rpb);
  return (long)nb * (Cout / 64) * (64 * 9 * 64);
}

// launcher used by focr_conv2d_wgrad (conv_igemm.hip); returns 1 if the layer was handled here
int focr_conv3x3_c64_wgrad(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats, int N,
                           int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx,
                           hipStream_t stream) {
  static const size_t lds = (size_t)12 * TILE_E * sizeof(__bf16);
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_wgrad_kernel_t<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_wgrad_kernel_t<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return 0;
    focr_dev_mark(attr_set);
  }
  if (const int ab = c3_wide_ab(N, H, W, Cin, Cout, KH, KW, padH, padW, ldd, ldx)) {
    int nb, rpb;
    c3_wide_blocks(N, H, Cin, Cout, ab, nb, rpb);
    const int ntco = Cout / 64, ntci = Cin / 64;
    if (!ws || ws_floats < (long)nb * ntco * ntci * (64 * 9 * 64)) return 0;      // the wide kernel of conv_bx3.hip takes it
    hipLaunchKernelGGL(conv3x3_c64_wgrad_kernel_t<true>, dim3(ntco, nb, ntci), 256, lds, stream, x, dy, dw, dbias, ws, N, H, W,
                       Cout, ldx, ldd, rpb, ab);
    const long n4 = (long)Cout * 9 * Cin / 4;
    // (dW arrives zeroed -- prezeroed or memset by the caller -- and the other paths ADD into it: so does this one)
    hipLaunchKernelGGL(conv3x3_wide_reduce_kernel, dim3((int)((n4 + 255) / 256)), 256, 0, stream, (const float*)ws, dw, ntco,
                       ntci, nb, 1);
    return 1;
  }
  if (!c3_applicable(W, Cin, Cout, KH, KW, padH, padW, ldd, ldx)) return 0;
  int nb, rpb;
  c3_blocks(N, H, Cout, nb, rpb);
  const int ntile = Cout / 64;
  float* part = (ws && ws_floats >= (long)nb * ntile * (64 * 9 * 64)) ? ws : nullptr;
  hipLaunchKernelGGL(conv3x3_c64_wgrad_kernel_t<false>, dim3(ntile, nb), 256, lds, stream, x, dy, dw, dbias, part, N, H, W,
                     Cout, ldx, ldd, rpb, 1);
  if (part)
    hipLaunchKernelGGL(conv3x3_c64_reduce_kernel, dim3(ntile * (64 * 9 * 64 / 4) / 8), 256, 0, stream, part, dw, ntile, nb);
  return 1;
}

// 3x3 / pad 1 NHWC convolution with the input tile RESIDENT in LDS ("halo" kernel), bf16 MFMA pipe.
//
// Replaces the generic implicit-GEMM kernel (conv_bx3.hip) for every 3x3 layer with Cin % 64 == 0 and
// Cout % 64 == 0: TBSRN/TSRN SRB conv1/conv2 and block7 (tbsrn.py:232-251, tsrn.py:77-98), the upsample conv
// (tsrn.py:104-114), the CRNN body (crnn.py:31-63) and -- on flipped weights -- their data gradients.
//
// Why: the generic kernel re-gathers every input element once per tap (9 x through L1/L2, 1.6 x measured HBM
// traffic), re-splits it to bf16 hi/lo in every consumer and needs two barriers per 32-deep K chunk.  Here
//   * a block owns 4 output rows x 32 pixels; the 6 x 34 pixel x 64 channel input halo is fetched ONCE, split to
//     bf16 hi/lo ONCE and kept in LDS (52 KB) while the nine taps are contracted from it; input-channel slices of 64
//     are staged one after the other for wider layers (accumulators persist);
//   * the weights arrive PRE-SPLIT and in MFMA-fragment order (weight_prep_frag_kernel below, one batched launch per
//     step), so a K chunk is a handful of contiguous 1 KB pieces that each wave DMAs straight into LDS
//     (global_load_lds, no VGPRs, no VALU); two chunk buffers, ONE barrier per 48-deep chunk (18 MFMAs per wave);
//   * every LDS fragment read is a conflict-free ds_read_b128: the halo's 16-byte channel chunks are XOR-swizzled by
//     the pixel index, the weight pieces are lane-linear by construction;
//   * two blocks (8 waves) per CU: one block's staging overlaps the other's MFMA stream;
//   * tiles of one image are consecutive on one XCD (block id % 8), so halo overlaps hit that XCD's L2;
//   * optional epilogue output: per-block BatchNorm partial sums (sum, sum of squares per channel), folded in a
//     fixed order by bn_fold (no separate statistics pass over the conv output).
//
// PLANES = 2: split products a_hi.b_hi + a_hi.b_lo + a_lo.b_hi (fp32-equivalent, forward);
// PLANES = 1: single bf16 product (data gradients under precision mode 3).
#include "focr_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 hbf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 hbf16x4;

#define H3_TR 4                                    // output rows per block (default shape)
#define H3_TP 32                                   // output pixels per row and block (default shape)
#define H3_HR (H3_TR + 2)
#define H3_HP (H3_TP + 2)
#define H3_PLANE_BYTES (H3_HR * H3_HP * 128)       // one bf16 plane of the halo: 26112 B (the largest of the shapes below)
// Tile shapes (round 6).  A block always owns 128 output pixels = 4 waves x one 32-pixel MFMA M tile, but maps narrower
// than 32 pixels wasted the rest of a 4 x 32 tile on padding -- half of it at W = 16, three quarters at W = 8, which is
// where the >= 256-channel layers of the SLD and text-focus ResNets live (16 x 16 and 8 x 8 maps).  TP = pixels per tile
// row: 32 (4 rows), 16 (8 rows: a wave's M tile = 2 rows x 16 px) or 8 (16 rows: 4 rows x 8 px); the halo is
// (128 / TP + 2) x (TP + 2) pixels = 204 / 180 / 180, i.e. never larger than the default's, so the LDS layout (plane
// offsets, weight buffers behind the halo) does not depend on the shape.
__host__ __device__ constexpr int h3_tp_for(int W) { return W <= 8 ? 8 : (W <= 16 ? 16 : 32); }
#define H3_KSC 2                                   // MFMA k-steps (16 channels) per weight chunk
#define H3_NCHUNK 18                               // 9 taps x 4 k-steps / 2
#ifndef H3_DIST
#define H3_DIST 2                                  // weight chunks in flight ahead of the one being contracted (two planes)
#endif
#ifndef H3_DIST1
#define H3_DIST1 5                                 // ... single-product launches (round 6)
#endif
// Weight chunk buffers: chunk c + DIST is in flight while c is contracted.  The buffer index of a chunk is (c % NBUF) with c
// counted per slice, so NBUF must divide the 18 chunks of a slice (3, 6, 9).  Two planes: 3 buffers of 8 KB -- deeper costs
// the second resident block (76.8 -> 101 KB of LDS) and loses (437 -> 524 us on 512 -> 512, B = 128).  One plane: chunks are
// 4 KB and the launch moves a third of the MFMAs per weight byte, so the DMA latency shows: 6 buffers (50.7 KB, still two
// blocks per CU).
template <int PLANES> struct H3Cfg {
  static constexpr int DIST = PLANES == 1 ? H3_DIST1 : H3_DIST;
  static constexpr int NBUF = DIST + 1;
  static_assert(H3_NCHUNK % NBUF == 0, "chunk buffers must divide the chunks of a slice");
};

// byte offset (inside one halo plane) of the 16-byte chunk c (channels 8c..8c+7) of halo pixel (r, p):
// a pixel is 128 B; the chunk index is XORed with bits 1..3 of p so that the 16 lanes of a ds_read_b128 group
// (16 different pixels, same logical chunk) land on 16 different 16-byte slots of the 256-byte bank row
__host__ __device__ inline int h3_off(int r, int p, int c, int hp = H3_HP) { return ((r * hp + p) * 8 + (c ^ ((p >> 1) & 7))) * 16; }

// Fragment-ordered split weights: element (row n, contraction index k) of an [Nrows][K] matrix, plane 0 = hi, 1 = lo.
// Piece (n / 32, k / 16, plane) is 1 KB: lane l = 32 * ((k >> 3) & 1) + (n & 31) holds k & 7 -- exactly the
// B operand of v_mfma_f32_32x32x16_bf16, so one wave-wide 16-byte access per lane moves a whole fragment.
__host__ __device__ inline size_t wfrag_index(int n, int k, int ksteps, int plane) {
  return ((((size_t)(n >> 5) * ksteps + (k >> 4)) * 2 + plane) * 64 + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7);
}

struct WPrepDesc {
  const float* w;      // [Cout][KH][KW][Cin] fp32 (OHWI)
  __bf16* wf;          // fragment-ordered output
  int cout, kh, kw, cin;
  int flip;            // 0: rows = Cout, k = (kh, kw, ci);  1 (data gradient): rows = Cin, k = (KH-1-kh, KW-1-kw, co)
  int pad_;
};

__host__ __device__ inline int wprep_rows(const WPrepDesc& d) { return d.flip ? d.cin : d.cout; }
__host__ __device__ inline int wprep_k(const WPrepDesc& d) { return d.kh * d.kw * (d.flip ? d.cout : d.cin); }
// bf16 elements of the fragment-ordered buffer (rows padded to 32, k to 16)
__host__ __device__ inline size_t wprep_elems(int rows, int k) {
  return (size_t)((rows + 31) / 32) * ((k + 15) / 16) * 2 * 64 * 8;
}

// one thread per (row, 8-k group): 8 source values -> one 16-byte store per plane
__device__ __forceinline__ void weight_prep_frag_body(const WPrepDesc& d) {
  const int rows = wprep_rows(d), K = wprep_k(d);
  const int rows_p = (rows + 31) / 32 * 32, ksteps = (K + 15) / 16;
  const int kc = d.flip ? d.cout : d.cin;            // innermost extent of k
  const long total = (long)rows_p * ksteps * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 31);
    long t = i >> 5;
    const int lh = (int)(t & 1);
    t >>= 1;
    const int ks = (int)(t % ksteps), nt = (int)(t / ksteps);
    const int n = nt * 32 + j;
    hbf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ks * 16 + lh * 8 + e;
      float v = 0.f;
      if (n < rows && k < K) {
        const int tap = k / kc, c = k - tap * kc;
        const int kh = tap / d.kw, kw = tap - kh * d.kw;
        if (d.flip)
          v = d.w[((size_t)(c * d.kh + (d.kh - 1 - kh)) * d.kw + (d.kw - 1 - kw)) * d.cin + n];
        else
          v = d.w[((size_t)(n * d.kh + kh) * d.kw + kw) * d.cin + c];
      }
      const __bf16 h = (__bf16)v;
      hi[e] = h;
      lo[e] = (__bf16)(v - (float)h);
    }
    const size_t o = (((size_t)nt * ksteps + ks) * 2) * 512 + (size_t)(lh * 32 + j) * 8;
    *reinterpret_cast<hbf16x8*>(d.wf + o) = hi;
    *reinterpret_cast<hbf16x8*>(d.wf + o + 512) = lo;
  }
}
__global__ __launch_bounds__(256) void weight_prep_frag_kernel(const WPrepDesc* __restrict__ descs) {
  const WPrepDesc d = descs[blockIdx.y];
  weight_prep_frag_body(d);
}
__global__ __launch_bounds__(256) void weight_prep_frag_one_kernel(WPrepDesc d) { weight_prep_frag_body(d); }

__device__ __forceinline__ void h3_split4(float4 v, hbf16x4& hi, hbf16x4& lo) { focr_split4(v, hi, lo); }

// ---- main-loop fragment traffic: hand-ordered ds_read_b128 + counted waits -----------------------------------
// hipcc's own schedule of this loop serialises: it re-uses two or three fragment registers and waits lgkmcnt(0) in
// front of every MFMA triple, so the LDS latency is exposed 36 times per tile.  The reads are therefore issued from
// inline asm in a fixed software pipeline (two fragment sets; the reads of step j + 1 are in flight while the six
// MFMAs of step j run) with counted s_waitcnt.  The compiler does not count asm loads: every wait statement names
// the registers it guards as "+v", which orders the MFMA builtins behind it (cdna_hip_programming.md 5.7 form ii).
typedef __attribute__((ext_vector_type(4))) int h3_i32x4;
template <int PLANES>
struct H3Frags {
  h3_i32x4 a[2];        // [plane]
  h3_i32x4 b[2][2];     // [column tile][plane]
};
template <int OFF>
__device__ __forceinline__ void h3_ldsr(h3_i32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF) : "memory");
}
// reads of local k-step J of chunk C (tap = (3C + J) >> 2): A from the halo, B from chunk buffer C & 1
template <int PLANES, int C, int J, int HP = H3_HP>
__device__ __forceinline__ void h3_read_step(H3Frags<PLANES>& f, const unsigned (&aoff)[3][4], unsigned boff) {
  constexpr int l = H3_KSC * C + J, tap = l >> 2, kk = l & 3, kh = tap / 3, kw = tap - 3 * kh;
  constexpr int ao = kh * (HP * 128);
  constexpr int bo = (C % H3Cfg<PLANES>::NBUF) * (2 * H3_KSC * PLANES * 1024);
  h3_ldsr<ao>(f.a[0], aoff[kw][kk]);
  if (PLANES == 2) h3_ldsr<ao + H3_PLANE_BYTES>(f.a[1], aoff[kw][kk]);
  h3_ldsr<bo + ((0 * H3_KSC + J) * PLANES + 0) * 1024>(f.b[0][0], boff);
  if (PLANES == 2) h3_ldsr<bo + ((0 * H3_KSC + J) * PLANES + 1) * 1024>(f.b[0][1], boff);
  h3_ldsr<bo + ((1 * H3_KSC + J) * PLANES + 0) * 1024>(f.b[1][0], boff);
  if (PLANES == 2) h3_ldsr<bo + ((1 * H3_KSC + J) * PLANES + 1) * 1024>(f.b[1][1], boff);
}
// wait until at most N of this wave's LDS reads are outstanding; f is the set that must be complete afterwards
template <int PLANES, int N>
__device__ __forceinline__ void h3_wait(H3Frags<PLANES>& f) {
  if (PLANES == 2)
    asm volatile("s_waitcnt lgkmcnt(%6)"
                 : "+v"(f.a[0]), "+v"(f.a[1]), "+v"(f.b[0][0]), "+v"(f.b[0][1]), "+v"(f.b[1][0]), "+v"(f.b[1][1])
                 : "i"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(f.a[0]), "+v"(f.b[0][0]), "+v"(f.b[1][0]) : "i"(N));
#ifndef H3_NO_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int PLANES>
__device__ __forceinline__ void h3_mfma_step(f32x16 (&acc)[2], const H3Frags<PLANES>& f) {
#ifdef H3_ABL_MFMA
  asm volatile("" ::"v"(f.a[0]), "v"(f.b[0][0]), "v"(f.b[1][0]));
  if (PLANES == 2) asm volatile("" ::"v"(f.a[1]), "v"(f.b[0][1]), "v"(f.b[1][1]));
  return;
#endif
  const hbf16x8 ah = __builtin_bit_cast(hbf16x8, f.a[0]);
  // (alternating the two accumulators instead of three dependent products in a row: no change, profiles/r06_halo_interleave_ab.txt)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const hbf16x8 bh = __builtin_bit_cast(hbf16x8, f.b[nt][0]);
    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nt], 0, 0, 0);
    if (PLANES == 2) {
      const hbf16x8 al = __builtin_bit_cast(hbf16x8, f.a[1]);
      const hbf16x8 bl = __builtin_bit_cast(hbf16x8, f.b[nt][1]);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nt], 0, 0, 0);
    }
  }
#ifndef H3_NO_PIN
  __builtin_amdgcn_sched_barrier(0);
#endif
}

struct H3Geom {
  int N, H, W, Cin, Cout, ldx, ldy, ldr;
  int tiles_x, tiles_y, cg_loop;   // cg_loop: 64-channel output groups walked inside one block (Cin == 64 only)
  int ksteps_total;                // 9 * Cin / 16
  int tiles, gfast;                // gfast: 1-D grid, the output groups of one tile are neighbours on one XCD (see the decode)
  int ldm;                         // pixel pitch of the mask source (epilogue: output kept where mask > 0, else 0)
};

#ifdef H3_TRACE
// tools/ubench/conv_ubench.cpp trace: per block [t_start, t_staged, t_contracted, t_end] (100 MHz wall clock), HW_ID, XCC_ID
__device__ unsigned long long* h3_trace_buf;
#define H3_STAMP(k) do { if (h3_trace_buf && threadIdx.x == 0) h3_trace_buf[(size_t)blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define H3_STAMP(k)
#endif

// MASK is a template parameter, not a run-time test: with `if (Mk)` around the eight mask registers EVERY instantiation
// spilled (private segment 20-100 -> 336-420 B, 8 -> 86-105 spilled VGPRs; the 64 -> 64 layer 43 -> 55 us alone,
// profiles/r06f_*) -- the unmasked kernel is the round-5 kernel, register for register.
// RES: the launch has a residual operand.  Launches without one -- most of the SR network's -- take the instantiation
// that does not carry its eight float4 per lane at all (161-179 VGPRs): 64 -> 64 at B = 128 42.6 -> 37.4 us,
// c3 -1.6 %, c5 -2.2 %, c1 -2.5 % (profiles/r06_halo_nores_ab.txt).
// (three blocks per CU for the single-product, no-residual instantiation -- 161 VGPRs, 50.7 KB of LDS, waves_per_eu(2, 3) --
// measured: no change, profiles/r06_halo_occ3_ab.txt)
template <int PLANES, int TP = H3_TP, bool MASK = false, bool RES = true>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_halo_kernel(const float* __restrict__ X, const __bf16* __restrict__ Wf,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ R, float* __restrict__ Y,
                                                              float* __restrict__ stats, H3Geom g, float alpha,
                                                              int relu, const float* __restrict__ Mk) {
  // Mk (round 6, data-gradient launches): the layer's INPUT x = relu(...) of the forward pass.  The gradient this launch
  // produces is the gradient of that relu's output; keeping it only where x > 0 IS the producer's relu backward, so the
  // producer skips its own pass (kernels.StepContext.premasked) -- one launch and one read + write of the tensor less per layer.
  constexpr int TR = 128 / TP, HR = TR + 2, HP = TP + 2;      // tile rows, halo rows / pixels per halo row
  constexpr int RPW = 32 / TP;                     // image rows of a wave's 32-pixel M tile
  constexpr int NP = 2 * H3_KSC * PLANES;          // 1 KB weight pieces per chunk (2 column tiles x 3 k-steps x planes)
  constexpr int BBUF = NP * 1024;
  constexpr int DIST = H3Cfg<PLANES>::DIST, NBUF = H3Cfg<PLANES>::NBUF;
  constexpr int HALO = PLANES * H3_PLANE_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char h3_smem[];   // [halo planes][B buf 0][B buf 1]
  unsigned char* const bbase = h3_smem + HALO;

  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  H3_STAMP(0);
#ifdef H3_TRACE
  if (h3_trace_buf && threadIdx.x == 0) {
    h3_trace_buf[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    h3_trace_buf[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
#ifdef H3_STAGGER
  // experiment (tools/gpu/r03_call19.sh): delay the second resident block of every CU (ids t and t + 32 of an XCD share a
  // CU, see the trace) so that its staging overlaps the first one's contraction.  No gain at any delay: flat up to
  // 3.4 us, then slower by the delay.
  // (round 6, tools/gpu/r06_call40.sh: the same on the multi-slice layers, linear block id)
  if ((((blockIdx.x + gridDim.x * blockIdx.y) >> 3) >> 5) & 1)
    for (int i = 0; i < H3_STAGGER; ++i) __builtin_amdgcn_s_sleep(16);
#endif
  // tile decode: the tiles of one image share halo rows/columns -> keep them on one XCD (block id % 8)
  // Multi-slice layers (round 6, gfast): grid = tiles x groups in ONE dimension, and inside an XCD's share of it the
  // output-channel group runs fastest -- the blocks that contract the SAME input tile against different weight panels are
  // resident together on one XCD, so the tile's halo comes out of HBM / Infinity Cache once and is an L2 hit for the other
  // groups (with the group in blockIdx.y the 2 x tiles resident blocks covered two groups, and every tile was fetched
  // groups / 2 times at different moments of the launch).  The slice's weights of all groups (9 x 64 x Cout x 2 planes:
  // 1.2 MB at Cout = 512) fit the XCD's L2 next to them.  Single-product launches with Cin >= 128 only: 512 -> 512 on
  // 8 x 32 maps at B = 128 215 -> 202 us, 512 -> 1024 407 -> 382, 128 -> 128 on 16 x 64 70.5 -> 65.8; the split-product
  // launches, three times the MFMAs per staged byte, do not wait for the halo and lose 0-3 % (profiles/r06_halo_group_fast_ab.txt).
  const int tpi = g.tiles_x * g.tiles_y;
  int img, tile, id, cgb;
  if (g.gfast) {
    const int ngr = g.Cout >> 6, L = blockIdx.x, j = L >> 3;
    cgb = j % ngr;
    id = (j / ngr) * 8 + (L & 7);
  } else {
    id = blockIdx.x;
    cgb = blockIdx.y;
  }
  {
    if ((g.N & 7) == 0) {
      const int xcd = id & 7, t = id >> 3;
      tile = t % tpi;
      img = (t / tpi) * 8 + xcd;
    } else {
      img = id / tpi;
      tile = id - img * tpi;
    }
  }
  const int y0 = (tile / g.tiles_x) * TR, x0 = (tile % g.tiles_x) * TP;
  const int nslices = g.Cin >> 6;
  const int ks_per_tap = g.Cin >> 4;

  // ---- weight chunk DMA: chunk c of slice s = k-steps l = 3c..3c+2 of the slice's 36, l -> tap l >> 2, step l & 3
  // (the per-lane source pointer is wl + a wave-uniform element offset: one 64-bit add per piece)
  const __bf16* const wl = Wf + lane * 8;
  auto issue_chunk = [&](int cg, int s, int c, int buf) {
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += 4) {
      const int p = p0 + wave;
      if (NP % 4 == 0 || p < NP) {
        const int pl = p % PLANES, t = p / PLANES, j = t % H3_KSC, nt = t / H3_KSC;
        const int l = H3_KSC * c + j;
        const int gks = (l >> 2) * ks_per_tap + 4 * s + (l & 3);
        const unsigned off = ((unsigned)((cg * 2 + nt) * g.ksteps_total + gks) * 2 + pl) * 512u;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wl + off),
                                         (__attribute__((address_space(3))) void*)(bbase + buf * BBUF + p * 1024), 16,
                                         0, 0);
      }
    }
  };

  // ---- halo staging: 204 pixels x 16 float4; thread (pixel = idx >> 4, c4 = idx & 15), two batches of loads
  auto stage_batch = [&](int s, auto i0c, auto i1c) {
    constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
    float4 v[I1 - I0];
    int tid = threadIdx.x;                      // opaque copy: keeps the per-pass index math out of the main loop's
    asm volatile("" : "+v"(tid));               // live ranges (hoisted as slice-invariant it costs ~50 VGPRs)
    // unconditional loads (out-of-image positions read the image's first pixel and are zeroed afterwards): a load
    // under a per-element branch makes hipcc wait for each one separately
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int idx = i * 256 + tid, pxl = idx >> 4, c4 = idx & 15;
      const int r = pxl / HP, p = pxl - r * HP;
      const int iy = y0 + r - 1, ix = x0 + p - 1;
      const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
      const int pix = ok ? iy * g.W + ix : 0;
      v[i - I0] = *reinterpret_cast<const float4*>(X + ((size_t)img * g.H * g.W + pix) * g.ldx + s * 64 + c4 * 4);
    }
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int idx = i * 256 + tid, pxl = idx >> 4, c4 = idx & 15;
      const int r = pxl / HP, p = pxl - r * HP;
      const int iy = y0 + r - 1, ix = x0 + p - 1;
      float4 x = v[i - I0];
      if (!((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)) x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pxl < HR * HP) {
        const int o = h3_off(r, p, c4 >> 1, HP) + (c4 & 1) * 8;
        if (PLANES == 2) {
          hbf16x4 h, l;
          h3_split4(x, h, l);
          *reinterpret_cast<hbf16x4*>(h3_smem + o) = h;
          *reinterpret_cast<hbf16x4*>(h3_smem + H3_PLANE_BYTES + o) = l;
        } else {
          hbf16x4 h;
          h[0] = (__bf16)x.x; h[1] = (__bf16)x.y; h[2] = (__bf16)x.z; h[3] = (__bf16)x.w;
          *reinterpret_cast<hbf16x4*>(h3_smem + o) = h;
        }
      }
    }
  };
  auto stage_halo = [&](int s) {
    constexpr int NPASS = (HR * HP * 16 + 255) / 256;           // 13 (12 for the narrow shapes)
    // round 3: two batches (7 + 6 loads per thread), one batch of 13 was SLOWER (staging phase 6.8 -> 7.4 us in the block
    // trace, launch 40.8 -> 42.0 us) -- with 31 spilled VGPRs in the split-product variant
    // (round 6, the kernel no longer spills: split-product launches take all 13 loads in one batch -- 512 -> 512 at B = 128
    // 421 -> 414 us, tfl -0.8 %; single-product launches are FASTER with 7 + 6: 174 vs 187 us, profiles/r06_halo_one_batch_ab.txt)
    if (PLANES == 2) {
      stage_batch(s, std::integral_constant<int, 0>{}, std::integral_constant<int, NPASS>{});
    } else {
      stage_batch(s, std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{});
      stage_batch(s, std::integral_constant<int, 7>{}, std::integral_constant<int, NPASS>{});
    }
  };

  // per-lane LDS byte addresses: A fragment of pixel (wave + kh, li + kw), channel chunk 2 * kk + lh (the tap row kh and
  // the plane are immediate offsets); B fragments are lane-linear inside their 1 KB piece
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)h3_smem;
  unsigned aoff[3][4];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      aoff[kw][kk] = lds0 + ((wave * RPW + li / TP) * HP + (li % TP) + kw) * 128 +
                     (((2 * kk + lh) ^ ((((li % TP) + kw) >> 1) & 7)) << 4);
  const unsigned boff = lds0 + HALO + lane * 16;
  const int cg0 = cgb * g.cg_loop;
  constexpr int RS = 3 * PLANES;                    // LDS reads per k-step

  constexpr int NW = NP / 4;                        // DMA instructions per wave and chunk
#pragma unroll
  for (int c = 0; c < DIST; ++c) issue_chunk(cg0, 0, c, c);
  for (int gi = 0; gi < g.cg_loop; ++gi) {
    const int cg = cg0 + gi;
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // residual values of the transposed epilogue (lane = 4 channels of pixel 4 i + lane / 16): requested and consumed in the
    // epilogue (see there)
    // (default shape: a wave's pixel m = 0..31 is (row y0 + wave, column x0 + m); narrow shapes: RPW rows of TP pixels)
    const int ec4 = lane & 15, epq = lane >> 4, eco = cg * 64 + ec4 * 4, oy = y0 + wave * RPW;
    float4 rv[RES ? 8 : 1];
    for (int s = 0; s < nslices; ++s) {
#ifndef H3_ABL_STAGE
      if (gi == 0 || nslices > 1) stage_halo(s);    // Cin == 64: the halo stays for every output group
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // halo + first chunk of the slice visible
      H3_STAMP(1);
      const bool more = (s + 1 < nslices) || (gi + 1 < g.cg_loop);
      const int ncg = s + 1 < nslices ? cg : cg + 1, nsl = s + 1 < nslices ? s + 1 : 0;
      H3Frags<PLANES> fa, fb;                       // fa: step 0 of a chunk, fb: step 1
      h3_read_step<PLANES, 0, 0, HP>(fa, aoff, boff);
      // chunk C (buffer C % 3): [DMA chunk C+2] read fb<-step1 | wait fa, MFMA fa | wait fb | chunk C+1 landed (counted
      // vmcnt: the pieces of C+2 stay in flight), barrier | read fa<-step0 of chunk C+1 | MFMA fb
#define H3_CHUNK(C)                                                                              \
  {                                                                                              \
    if ((C) + DIST < H3_NCHUNK) issue_chunk(cg, s, (C) + DIST, ((C) + DIST) % NBUF);             \
    else if (more) issue_chunk(ncg, nsl, (C) + DIST - H3_NCHUNK, ((C) + DIST) % NBUF);           \
    h3_read_step<PLANES, (C), 1, HP>(fb, aoff, boff);                                               \
    h3_wait<PLANES, RS>(fa);                                                                     \
    h3_mfma_step<PLANES>(acc, fa);                                                               \
    h3_wait<PLANES, 0>(fb);                                                                      \
    if ((C) + DIST < H3_NCHUNK || more) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DIST - 1) * NW) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        \
    __builtin_amdgcn_s_barrier();                                                                \
    if ((C) + 1 < H3_NCHUNK) h3_read_step<PLANES, ((C) + 1) % H3_NCHUNK, 0, HP>(fa, aoff, boff);     \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    h3_mfma_step<PLANES>(acc, fb);                                                               \
  }
      H3_CHUNK(0) H3_CHUNK(1) H3_CHUNK(2) H3_CHUNK(3) H3_CHUNK(4) H3_CHUNK(5)
      H3_CHUNK(6) H3_CHUNK(7) H3_CHUNK(8) H3_CHUNK(9) H3_CHUNK(10) H3_CHUNK(11)
      H3_CHUNK(12) H3_CHUNK(13) H3_CHUNK(14) H3_CHUNK(15) H3_CHUNK(16) H3_CHUNK(17)
#undef H3_CHUNK
      H3_STAMP(2);
    }
    // ---- epilogue of this 64-channel output group
#ifdef H3_ABL_EPI
    if (acc[0][0] == 12345.678f && acc[1][3] == 0.5f) Y[tid] = acc[0][1];
    continue;
#endif
    if (g.cg_loop == 1) {
      // The halo is dead (one output group per block): every wave transposes its 32 pixel x 64 channel tile through
      // its own 8 KB of it, so that a lane owns 4 consecutive channels of one pixel: 16-byte residual loads and
      // output stores (a pixel's 256 B row is one contiguous access of 16 lanes) instead of 32 + 32 scalar ones
      float* const tw = reinterpret_cast<float*>(h3_smem) + wave * (32 * 64);
      // The residual is requested HERE, after the contraction (round 6).  Through round 5 it was requested before the last
      // slice's contraction and consumed after it: 32 registers live across the main loop of a kernel at the 256-VGPR limit
      // -- the source of every spill this kernel had (20-24 VGPRs, 84-100 B of scratch per thread = ~10 MB of writes per
      // launch).  Requested late the launch waits for it under the LDS transposition instead, and every instantiation is
      // spill-free: tfl 37.2 -> 36.6 ms, sfl 40.9 -> 40.3 (profiles/r06_halo_late_res_ab.txt).
      if (RES) {
        int x0r = x0, oyr = oy;
        asm volatile("" : "+v"(x0r), "+v"(oyr));
#pragma unroll
        for (int i = 0; i < (RES ? 8 : 0); ++i) {
          const int m = 4 * i + epq, ox = x0r + m % TP, oyy = oyr + m / TP;
          const bool ok = oyy < g.H && ox < g.W;
          const size_t pix = (size_t)(img * g.H + (ok ? oyy : 0)) * g.W + (ok ? ox : 0);
          rv[RES ? i : 0] = R ? *reinterpret_cast<const float4*>(R + pix * g.ldr + eco) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      float4 mv[MASK ? 8 : 1];                    // mask source values (requested here, under the transposition)
      if (MASK) {
        int x0m = x0, oym = oy;                   // opaque: these addresses are formed HERE, not hoisted above the main loop
        asm volatile("" : "+v"(x0m), "+v"(oym));
#pragma unroll
        for (int i = 0; i < (MASK ? 8 : 0); ++i) {
          const int m = 4 * i + epq, ox = x0m + m % TP, oyy = oym + m / TP;
          const bool ok = oyy < g.H && ox < g.W;
          const size_t pix = (size_t)(img * g.H + (ok ? oyy : 0)) * g.W + (ok ? ox : 0);
          mv[i] = *reinterpret_cast<const float4*>(Mk + pix * g.ldm + eco);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) tw[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + nt * 32 + li] = acc[nt][r];
      const int c4 = ec4, pq = epq, co = eco;
      int x0e = x0;                               // opaque: the store addresses are recomputed here instead of being
      asm volatile("" : "+v"(x0e));               // kept (spilled) across the main loop next to the residual's
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bv = *reinterpret_cast<const float4*>(bias + co);
      float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(tw + (4 * i + pq) * 64 + c4 * 4);
        const float4 rr = RES ? rv[RES ? i : 0] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = make_float4(alpha * a.x + bv.x + rr.x, alpha * a.y + bv.y + rr.y, alpha * a.z + bv.z + rr.z,
                               alpha * a.w + bv.w + rr.w);
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (MASK) {
          const float4 mk = mv[MASK ? i : 0];
          v = make_float4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f, mk.w > 0.f ? v.w : 0.f);
        }
        const int m = 4 * i + pq, oyy = oy + m / TP;
        if (oyy < g.H && x0 + m % TP < g.W) {
          const size_t pix = (size_t)(img * g.H + oyy) * g.W + x0e + m % TP;
          *reinterpret_cast<float4*>(Y + pix * g.ldy + co) = v;
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
        }
      }
      if (stats) {
        // per-block BatchNorm partial sums: fold the 4 pixel quarters of the wave (lanes c4 + 16 j), then the 4 waves
        // through the tail of the B buffers (the transposition tiles occupy the first 32 KB of the halo)
        float e[8] = {s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          e[k] += __shfl_xor(e[k], 16, 64);
          e[k] += __shfl_xor(e[k], 32, 64);
        }
        float* red = reinterpret_cast<float*>(bbase);            // [wave][64 ch][2]
        if (lane < 16) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            red[(wave * 64 + c4 * 4 + k) * 2] = e[k];
            red[(wave * 64 + c4 * 4 + k) * 2 + 1] = e[4 + k];
          }
        }
        __syncthreads();
        if (tid < 128) {
          const float v = red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];      // fixed order
          stats[((size_t)id * g.Cout + cg * 64 + (tid >> 1)) * 2 + (tid & 1)] = v;
        }
      }
    } else {
      // several output groups per block (Cin == 64, the halo must stay): direct stores; the residual values are
      // loaded as one batch from clamped addresses (never under a per-element branch)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int co = cg * 64 + nt * 32 + li;
        const float b = bias ? bias[co] : 0.f;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh, ox = x0 + m % TP, oyy = oy + m / TP;
          const bool ok = oyy < g.H && ox < g.W;
          const size_t pix = (size_t)(img * g.H + (ok ? oyy : 0)) * g.W + (ok ? ox : 0);
          rv[r] = (RES && R) ? R[pix * g.ldr + co] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lh, ox = x0 + m % TP, oyy = oy + m / TP;
          float v = alpha * acc[nt][r] + b + rv[r];
          if (relu) v = fmaxf(v, 0.f);
          if (oyy < g.H && ox < g.W) {
            const size_t pix = (size_t)(img * g.H + oyy) * g.W + ox;
            if (MASK && !(Mk[pix * g.ldm + co] > 0.f)) v = 0.f;
            Y[pix * g.ldy + co] = v;
          }
        }
      }
    }
  }
  H3_STAMP(3);
}

// ------------------------------------------------------------------------------------------------------------
// launchers (used by conv_igemm.hip's entry points and tools/ubench)
// ------------------------------------------------------------------------------------------------------------
int focr_conv3x3_halo_eligible(int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, int ldx) {
  return KH == 3 && KW == 3 && padH == 1 && padW == 1 && Cin % 64 == 0 && Cout % 64 == 0 && ldx % 4 == 0 && H >= 1 &&
         W >= 1;
}

template <int PLANES, int TP>
static int launch_h3(const float* x, const __bf16* wf, const float* bias, const float* residual, float* y, float* stats,
                     int N, int H, int W, int Cin, int Cout, int ldx, int ldy, int ldr, float alpha, int relu,
                     hipStream_t stream, const float* mask = nullptr, int ldm = 0) {
  constexpr int LDS = PLANES * H3_PLANE_BYTES + H3Cfg<PLANES>::NBUF * (2 * H3_KSC * PLANES) * 1024;
  constexpr int TR = 128 / TP;
  static focr_dev_flags attr_set;
  if (focr_dev_first(attr_set)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<PLANES, TP, false, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<PLANES, TP, false, false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<PLANES, TP, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return 0;
    focr_dev_mark(attr_set);
  }
  H3Geom g;
  g.N = N; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ldx = ldx; g.ldy = ldy; g.ldr = ldr;
  g.ldm = ldm;
  g.tiles_x = (W + TP - 1) / TP;
  g.tiles_y = (H + TR - 1) / TR;
  g.ksteps_total = 9 * Cin / 16;
  const int groups = Cout / 64;
  // Cin == 64: the halo is staged once and every output group is contracted from it inside the block -- unless that
  // leaves too few blocks to fill the chip
  const int tiles = N * g.tiles_x * g.tiles_y;
  g.cg_loop = (Cin == 64 && tiles >= 512 && !stats) ? groups : 1;   // the statistics epilogue reuses the halo's LDS
  static const bool gfast_on = !(getenv("FOCR_H3_GROUP_FAST") && getenv("FOCR_H3_GROUP_FAST")[0] == '0');
  g.tiles = tiles;
  g.gfast = (gfast_on && PLANES == 1 && Cin >= 128 && g.cg_loop == 1 && groups > 1 && tiles % 8 == 0 && (long)tiles * groups < (1l << 31)) ? 1 : 0;
  dim3 grid(g.gfast ? tiles * groups : tiles, g.gfast ? 1 : groups / g.cg_loop);
  static const bool nores_on = !(getenv("FOCR_H3_NORES") && getenv("FOCR_H3_NORES")[0] == '0');
  if (mask)
    hipLaunchKernelGGL((conv3x3_halo_kernel<PLANES, TP, true, true>), grid, 256, LDS, stream, x, wf, bias, residual, y, stats,
                       g, alpha, relu, mask);
  else if (!residual && nores_on)
    hipLaunchKernelGGL((conv3x3_halo_kernel<PLANES, TP, false, false>), grid, 256, LDS, stream, x, wf, bias, residual, y, stats,
                       g, alpha, relu, mask);
  else
    hipLaunchKernelGGL((conv3x3_halo_kernel<PLANES, TP, false, true>), grid, 256, LDS, stream, x, wf, bias, residual, y, stats,
                       g, alpha, relu, mask);
  return 1;
}
template <int PLANES>
static int launch_h3_shape(const float* x, const __bf16* wf, const float* bias, const float* residual, float* y, float* stats,
                           int N, int H, int W, int Cin, int Cout, int ldx, int ldy, int ldr, float alpha, int relu,
                           hipStream_t stream, const float* mask, int ldm) {
  switch (h3_tp_for(W)) {
    case 8: return launch_h3<PLANES, 8>(x, wf, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, stream, mask, ldm);
    case 16: return launch_h3<PLANES, 16>(x, wf, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, stream, mask, ldm);
    default: return launch_h3<PLANES, 32>(x, wf, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, stream, mask, ldm);
  }
}

// stats != nullptr: float [tiles][Cout][2] per-block (sum, sum of squares) of the stored outputs; returns the number
// of stat rows (tiles) through *stat_rows.  planes: 2 = split products, 1 = single bf16.
int focr_conv3x3_halo(const float* x, const void* wfrag, const float* bias, const float* residual, float* y,
                      float* stats, int N, int H, int W, int Cin, int Cout, int ldx, int ldy, int ldr, float alpha,
                      int relu, int planes, hipStream_t stream, const float* mask = nullptr, int ldm = 0) {
  const __bf16* wf = reinterpret_cast<const __bf16*>(wfrag);
  if (planes == 1)
    return launch_h3_shape<1>(x, wf, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, stream, mask, ldm);
  return launch_h3_shape<2>(x, wf, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, stream, mask, ldm);
}
int focr_conv3x3_halo_tiles(int N, int H, int W) {
  const int tp = h3_tp_for(W), tr = 128 / tp;
  return N * ((W + tp - 1) / tp) * ((H + tr - 1) / tr);
}

int focr_weight_prep_frag_launch(const void* descs_dev, int n, long max_threads, hipStream_t stream) {
  int gx = (int)((max_threads + 255) / 256);
  if (gx > 128) gx = 128;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(weight_prep_frag_kernel, dim3(gx, n), 256, 0, stream,
                     reinterpret_cast<const WPrepDesc*>(descs_dev));
  return 1;
}

// ------------------------------------------------------------------------------------------------------------
// C ABI (include/focr.h)
// ------------------------------------------------------------------------------------------------------------
extern "C" long focr_weight_frag_bytes(int rows, int k) { return (long)wprep_elems(rows, k) * 2; }

extern "C" int focr_weight_prep_frag(const float* w, void* wfrag, int Cout, int KH, int KW, int Cin, int flip,
                                     hipStream_t stream) {
  FOCR_CHECK_ARG(w && wfrag && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "bad argument");
  WPrepDesc d{w, reinterpret_cast<__bf16*>(wfrag), Cout, KH, KW, Cin, flip ? 1 : 0, 0};
  const long threads = (long)wprep_elems(wprep_rows(d), wprep_k(d)) / 16;
  int gx = (int)((threads + 255) / 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(weight_prep_frag_one_kernel, dim3(gx), 256, 0, stream, d);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_weight_prep_frag_batched(const void* descs_dev, int n, long max_threads, hipStream_t stream) {
  FOCR_CHECK_ARG(descs_dev && n > 0 && max_threads > 0, "bad argument");
  focr_weight_prep_frag_launch(descs_dev, n, max_threads, stream);
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_conv3x3_frag_tiles(int N, int H, int W) { return focr_conv3x3_halo_tiles(N, H, W); }

static int conv3x3_frag_fwd_impl(const float* x, const void* wfrag, const float* bias, const float* residual,
                                     float* y, float* stats, int N, int H, int W, int Cin, int Cout, float alpha,
                                     int relu, int planes, int ldy, int ldr, int ldx, const float* mask, int ldm,
                                     hipStream_t stream) {
  FOCR_CHECK_ARG(x && wfrag && y && N > 0, "null pointer");
  if (ldy <= 0) ldy = Cout;
  if (ldr <= 0) ldr = Cout;
  if (ldx <= 0) ldx = Cin;
  FOCR_CHECK_ARG(focr_conv3x3_halo_eligible(H, W, Cin, Cout, 3, 3, 1, 1, ldx), "layer shape not supported");
  FOCR_CHECK_ARG(ldy % 4 == 0 && ldr % 4 == 0 && ldy >= Cout && ldr >= Cout && ldx >= Cin, "bad row pitch");
  FOCR_CHECK_ARG(planes == 1 || planes == 2, "planes must be 1 or 2");
  if (mask && ldm <= 0) ldm = Cout;
  FOCR_CHECK_ARG(!mask || (ldm % 4 == 0 && ldm >= Cout && (reinterpret_cast<uintptr_t>(mask) & 15) == 0), "bad mask pitch / alignment");
  if (!focr_conv3x3_halo(x, wfrag, bias, residual, y, stats, N, H, W, Cin, Cout, ldx, ldy, ldr, alpha, relu, planes,
                         stream, mask, ldm)) {
    focr_set_error("focr_conv3x3_frag_fwd: launch setup failed");
    return FOCR_EHIP;
  }
  FOCR_LAUNCH_CHECK();
  return FOCR_OK;
}

extern "C" int focr_conv3x3_frag_fwd(const float* x, const void* wfrag, const float* bias, const float* residual,
                                     float* y, float* stats, int N, int H, int W, int Cin, int Cout, float alpha,
                                     int relu, int planes, int ldy, int ldr, int ldx, hipStream_t stream) {
  return conv3x3_frag_fwd_impl(x, wfrag, bias, residual, y, stats, N, H, W, Cin, Cout, alpha, relu, planes, ldy, ldr, ldx,
                               nullptr, 0, stream);
}
// as focr_conv3x3_frag_fwd, and the output is kept only where mask[pixel][channel] > 0 (mask: [N][H][W][ldm >= Cout] fp32):
// the relu backward of the layer that produced this launch's input, applied to the data gradient it produces
extern "C" int focr_conv3x3_frag_fwd_masked(const float* x, const void* wfrag, const float* bias, const float* residual,
                                            float* y, int N, int H, int W, int Cin, int Cout, float alpha, int planes,
                                            int ldy, int ldr, int ldx, const float* mask, int ldm, hipStream_t stream) {
  FOCR_CHECK_ARG(mask, "null mask");
  return conv3x3_frag_fwd_impl(x, wfrag, bias, residual, y, nullptr, N, H, W, Cin, Cout, alpha, 0, planes, ldy, ldr, ldx, mask,
                               ldm, stream);
}

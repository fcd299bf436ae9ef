// Shared helpers for the libfocr_hip C-ABI kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define FOCR_OK 0
#define FOCR_EINVAL (-1)
#define FOCR_EUNSUPPORTED (-2)
#define FOCR_EHIP (-3)
#define FOCR_ENCCL (-4)

extern "C" void focr_set_error(const char* fmt, ...);
extern "C" int focr_get_precision(void);
// A/B kernel-selection switches (focr_core.hip focr_set_tuning)
#define FOCR_TUNE_LINEAR_WGRAD_STREAM 0
#define FOCR_TUNE_ATTN_FWD_VARIANT 1
#define FOCR_TUNE_LSTM_PERSISTENT 2
#define FOCR_TUNE_ATTN_BWD_DQ_VARIANT 3
#define FOCR_TUNE_ATTN_FWD_MASK 4
#define FOCR_TUNE_GRU_LOADER 5
#define FOCR_TUNING_COUNT 6
extern "C" int focr_get_tuning(int key);

#define FOCR_CHECK_ARG(cond, msg)                          \
  do {                                                     \
    if (!(cond)) {                                         \
      focr_set_error("%s: %s", __func__, msg);             \
      return FOCR_EINVAL;                                  \
    }                                                      \
  } while (0)

#define FOCR_LAUNCH_CHECK()                                              \
  do {                                                                   \
    hipError_t e_ = hipGetLastError();                                   \
    if (e_ != hipSuccess) {                                              \
      focr_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_)); \
      return FOCR_EHIP;                                                  \
    }                                                                    \
  } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 focr_bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 focr_bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 focr_bf16x8;

// x = hi + lo split of TWO floats at once: 6 VALU instructions per pair (v_cvt_pk_bf16_f32, shift, and, 2 x v_sub_f32,
// v_cvt_pk_bf16_f32).  Written on pairs because hipcc otherwise converts the hi part twice (once alone to form x - hi,
// once as a pair for packing): 8 instructions per pair.  The two subtractions stay scalar on purpose: beside MFMAs a
// v_pk_add_f32 costs more than the two v_sub_f32 it replaces (MI355X_MICROARCH.md, "price of one filler"; measured here:
// tools/gpu/r03_call22.sh vs r03_call23.sh).  Same values as the scalar form (round-to-nearest-even conversion, exact
// widening, exact subtraction).
__device__ __forceinline__ void focr_split2(f32x2 x, focr_bf16x2& hi, focr_bf16x2& lo) {
  // no fma contraction across the subtraction: when x is a product (x = v * scale), lo must come from the ROUNDED product
  // that hi was taken from, so that every producer of split operands (kernels splitting on the fly, attn_make_planes,
  // fe_qkv_fwd writing planes) yields the same bits
#pragma clang fp contract(off)
  hi = __builtin_convertvector(x, focr_bf16x2);
  const unsigned w = __builtin_bit_cast(unsigned, hi);
  float dx = x.x - __uint_as_float(w << 16);
  float dy = x.y - __uint_as_float(w & 0xffff0000u);
#ifndef FOCR_SPLIT_PK
  asm volatile("" : "+v"(dx));          // keeps the SLP vectoriser from fusing the pair into v_pk_add_f32 (a plain asm
                                        // measured equal in the default mode and 1.3 % slower in mode 1: r03_call32.sh)
#endif
  lo = __builtin_convertvector(f32x2{dx, dy}, focr_bf16x2);
}
__device__ __forceinline__ void focr_split4(float a, float b, float c, float d, focr_bf16x4& hi, focr_bf16x4& lo) {
  focr_bf16x2 h0, l0, h1, l1;
  focr_split2(f32x2{a, b}, h0, l0);
  focr_split2(f32x2{c, d}, h1, l1);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
}
__device__ __forceinline__ void focr_split4(float4 v, focr_bf16x4& hi, focr_bf16x4& lo) {
  focr_split4(v.x, v.y, v.z, v.w, hi, lo);
}
__device__ __forceinline__ void focr_split8(const float (&v)[8], focr_bf16x8& hi, focr_bf16x8& lo) {
  focr_bf16x4 h0, l0, h1, l1;
  focr_split4(v[0], v[1], v[2], v[3], h0, l0);
  focr_split4(v[4], v[5], v[6], v[7], h1, l1);
  hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
  lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// "done once" flags that are PER DEVICE: function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) and occupancy
// answers belong to a device, and a process may touch several (tests that switch device, DataParallel-style hosts).
#include <atomic>
struct focr_dev_flags { std::atomic<unsigned long long> bits{0ull}; };
static inline int focr_cur_device() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
static inline bool focr_dev_first(const focr_dev_flags& f) { return !(f.bits.load(std::memory_order_acquire) >> focr_cur_device() & 1ull); }
static inline void focr_dev_mark(focr_dev_flags& f) { f.bits.fetch_or(1ull << focr_cur_device(), std::memory_order_release); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// mish(x) = x * tanh(softplus(x)), softplus threshold 20 (reference tsrn.py:117-125), with ONE exp:
//   e = exp(x), w = e^2 + 2e  ->  tanh(log(1+e)) = w / (w + 2)   (exact identity; x > 20 -> softplus = x,
//   tanh(x) = 1 in fp32).  Replaces expf + log1pf + tanhf on every BN-mish / pixel-shuffle element.
// The quotients use v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions each): these functions
// sit in HBM-bound elementwise kernels that the divisions had made VALU-bound (pixel-shuffle backward 187 us for
// 400 MB).  Branch-free: e = exp(min(x, 20)) keeps w below 2.4e17, so w / (w + 2) is 1 to fp32 precision there.
__device__ __forceinline__ float mish_tanh_sp(float x) {
  const float e = __expf(fminf(x, 20.f));
  const float w = e * (e + 2.f);
  return w * __builtin_amdgcn_rcpf(w + 2.f);
}
__device__ __forceinline__ float mish_f(float x) { return x * mish_tanh_sp(x); }
// d mish / dx = t + x * (1 - t^2) * sigmoid(x)
//   with one reciprocal: t = w r, 1 - t^2 = 4 (w + 1) r^2, sigmoid = e / (1 + e), r = 1 / (w + 2), (1 + e)^2 = w + 1:
//   x (1 - t^2) sigmoid = 4 x e (1 + e) r^2
__device__ __forceinline__ float mish_grad_f(float x) {
  const float xc = fminf(x, 20.f);
  const float e = __expf(xc);
  const float w = e * (e + 2.f);
  const float r = __builtin_amdgcn_rcpf(w + 2.f);
  const float g = w * r + 4.f * xc * e * (1.f + e) * r * r;
  return x > 20.f ? 1.f : g;
}

// 32-bit two-level counter hash for the attention dropout mask (cheap: ~8 VALU ops per element):
// rowkey = rng_rowkey(seed, global query row) once per row, then rng_elem(rowkey, key).
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng_rowkey(uint64_t seed, uint32_t row) {
  return hash32(hash32(row ^ (uint32_t)seed) + (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint32_t rng_elem(uint32_t rowkey, uint32_t col) {
  return hash32(rowkey ^ (col * 0x9E3779B1U));
}
// Step epoch (focr_core.hip focr_set_seed_epoch): when the host has registered a device-resident step state, every
// dropout seed is folded with the state's epoch word ON THE DEVICE, so a launch whose scalar arguments never change (a
// replayed recording, replay.hip) still draws fresh keep bits in every step.  epoch == NULL: the seed as passed.
extern "C" const uint64_t* focr_seed_epoch(void);
__device__ __forceinline__ uint64_t focr_epoch_seed(uint64_t seed, const uint64_t* __restrict__ epoch) {
  return epoch ? seed + epoch[0] * 0xD1B54A32D192ED03ull : seed;
}
__device__ __forceinline__ uint32_t focr_epoch_seed32(uint32_t seed, const uint64_t* __restrict__ epoch) {
  return epoch ? seed ^ hash32((uint32_t)epoch[0] * 0x9E3779B1U + 0x85EBCA6BU) : seed;
}

// Attention block order: 1-D grid of BH * nqb blocks.  Workgroups are dispatched round-robin over the
// 8 XCDs (block id % 8), so ids {x, x+8, x+16, ...} share an XCD and its L2: give those consecutive
// ids to the nqb query/key blocks of ONE (batch, head), whose K/V (256 KB) then stays L2-resident.
__device__ __forceinline__ void attn_block_decode(int id, int BH, int nqb, int& bh, int& qb) {
  if ((BH & 7) == 0) {
    int xcd = id & 7, t = id >> 3;
    qb = t % nqb;
    bh = (t / nqb) * 8 + xcd;
  } else {
    bh = id % BH;
    qb = id / BH;
  }
}

// keep-bit application in 2 VALU ops: sign-extended 1-bit field (0 / 0xFFFFFFFF, v_bfe_i32) AND float bits
__device__ __forceinline__ float keep_if_bit(float v, uint32_t w, int bit) {
  int m = ((int)(w << (31 - bit))) >> 31;
  return __int_as_float(__float_as_int(v) & m);
}

// ---- attention dropout keep bits --------------------------------------------------------------------
// uint32 [B*H][Ntok/32 query groups][Ntok/32 key groups][32 slots]; bit j of a word = query 32*qg + j, the word's
// key is 32*kg + kk with slot = mask_slot(kk).  Slots are ordered so that the two keys held by accumulator
// register r of the S^T tile (lanes 0-31: key koff(r), lanes 32-63: key koff(r)+4, lane%32 = query) are the two
// halves of the 64-bit word r of the group: a ready-made lane mask for v_cndmask (forward / dQ kernels: ONE VALU
// op per score, the words arrive through scalar loads).  The dK/dV kernel (lane = key) loads its key's word
// per lane and tests bit (query offset) with v_bfe_i32.
// attention dropout probability, quantised to 1/4096 and expressed in 1/65536 units (low 4 bits zero): the keep-bit
// generator needs one xorshift word per significant bit of the keep probability, so 12-bit quantisation (p = 0.1 ->
// 0.10010) saves a third of its work; every kernel derives its 1/(1-p) from the same number
__host__ __device__ __forceinline__ uint32_t attn_drop_thr16(float p_drop) {
  return (uint32_t)(p_drop * 4096.0f + 0.5f) << 4;
}
__device__ __forceinline__ int mask_slot(int kk) { return 2 * ((kk & 3) + 4 * (kk >> 3)) + ((kk >> 2) & 1); }
// (builtin, not inline asm: the compiler must see the instruction to honour the trans-use / MFMA-read hazards)
__device__ __forceinline__ float keep_lanes(float x, uint64_t m) {
  return __builtin_amdgcn_inverse_ballot_w64(m) ? x : 0.f;
}
__device__ __forceinline__ int bit_sext(uint32_t w, int bit) { return __builtin_amdgcn_sbfe((int)w, bit, 1); }

// counter-based RNG for dropout masks: one 32-bit draw per (seed, index); the same
// function regenerates the mask in the backward kernels.
__device__ __forceinline__ uint32_t rng_hash(uint64_t seed, uint64_t idx) {
  uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

// library-wide state of the C ABI: thread-local error string, version.
#include "focr_common.h"
#include <stdarg.h>
#include <atomic>

static thread_local char g_err[512] = "";

extern "C" void focr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* focr_last_error(void) { return g_err; }
extern "C" int focr_version(void) { return 100; }

// contraction precision of the MFMA kernels that have both paths:
//   0 = exact fp32 on the f32-input MFMA (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak)
//   1 = split bf16 ("bf16x3": hi/lo operands, 3 products, fp32 accumulate) on v_mfma_f32_32x32x16_bf16
//   2 = as 1, but the three gradient ACCUMULATIONS of the attention backward (dV += P^T dO, dK += dS^T Q,
//       dQ += dS K) use single bf16 products: they have no softmax-style cancellation, their rounding error
//       (2^-9 per term, averaged over 1024 terms) is far below the 1e-2 gradient gate, while the score recompute
//       and dP = dO V^T (which feed exp / the dP - D cancellation) stay bf16x3.  Forward results are identical to 1.
//   3 = as 2, and the DATA-GRADIENT convolutions that run on the halo kernel (conv3x3_halo.hip) use a single bf16
//       product (activations' gradients and weights rounded to bf16, fp32 accumulate) -- the arithmetic of every
//       bf16 mixed-precision training stack.  Forward results are identical to 1 and 2; the host selects the plane
//       count per call (focr_conv3x3_frag_fwd), this flag only tells it what the library-wide choice is.
//       Round 4: the same rule for the attention's data gradient dP = dO V^T in the single-pass backward
//       (attention_bwd1_bx3.h, template flag DP1: dO and V rounded to bf16, one product; the score recomputation stays
//       split).  Measured against fp64 with the kernel's own keep bits: d qkv error 3.1e-4 .. 5.7e-4 of max (mode 2:
//       2.4e-4 .. 4.9e-4; gate 3e-3), kernel 444 -> 396 us, step 13.87 -> 13.55 ms.
// (atomics: both words are read from PyTorch autograd worker threads while the main thread may set them -- SURVEY 8(b):
// no unguarded global state.  Relaxed order is enough, a mode switch is only meaningful between steps.)
static std::atomic<int> g_precision{2};
extern "C" int focr_set_precision(int mode) {
  if (mode < 0 || mode > 3) {
    focr_set_error("focr_set_precision: mode must be 0 (fp32), 1 (bf16x3), 2 (bf16x3, bf16 gradient accumulation) or 3 (2 + bf16 data gradients)");
    return FOCR_EINVAL;
  }
  g_precision.store(mode, std::memory_order_relaxed);
  return FOCR_OK;
}
extern "C" int focr_get_precision(void) { return g_precision.load(std::memory_order_relaxed); }

// Kernel-selection switches for A/B measurements (tools/dev, tools/ubench): every switch has ONE production value (the
// default); results are the same either way, only the kernel that computes them changes.
//   0 "linear_wgrad_stream"  1: transformer-linear weight gradients on linear_wgrad.hip   0: generic split kernel
//   1 "attn_fwd_variant"     1: 256-query attention forward blocks    0: 128-query blocks    2: scores one key
//                            group ahead + thresholded rescale (measured: no gain in the step)
//   2 "lstm_persistent"      1: one launch per BiLSTM layer and direction pair (rnn.hip)  0: one launch per time step
//   3 "attn_bwd_dq_variant"  3: single pass with one wave per SIMD (attention_bwd1w_bx3.h; measured slower: A/B only)
//                            2: single-pass backward (dQ, dK, dV from one S / dP evaluation, attention_bwd1_bx3.h; precision
//                            modes 2 / 3, Ntok % 256 == 0, else as 1)   1: two passes, dQ pass with 256-query blocks (two
//                            tiles per wave)   0: two passes, 128-query blocks
//   5 "gru_loader"          2: 16-sequence compute waves on the 16x16x32 MFMA + one loader wave (default)
//                            1: TSRN GRU scans as loader / compute wave pairs (operands of the next steps DMA'd into an LDS
//                            ring by a second wave, rnn.hip)   0: single-wave scans with register prefetch (rounds 1-5)
static std::atomic<int> g_tuning[FOCR_TUNING_COUNT] = {{1}, {1}, {1}, {4}, {2}, {2}};
extern "C" int focr_set_tuning(int key, int value) {
  if (key < 0 || key >= FOCR_TUNING_COUNT) {
    focr_set_error("focr_set_tuning: unknown key %d", key);
    return FOCR_EINVAL;
  }
  g_tuning[key].store(value, std::memory_order_relaxed);
  return FOCR_OK;
}
extern "C" int focr_get_tuning(int key) { return (key >= 0 && key < FOCR_TUNING_COUNT) ? g_tuning[key].load(std::memory_order_relaxed) : -1; }

// ---- device-resident step state (include/focr.h focr_step_*): 64 bytes of caller-owned device memory
//   [0] u64 epoch   -- dropout epoch, folded into every dropout seed on the device (focr_common.h focr_epoch_seed)
//   [1] i64 t       -- optimiser step count (1-based after the first focr_step_advance)
//   [2] f32 c1, f32 c2s -- Adam bias corrections 1 - beta1^t and sqrt(1 - beta2^t) of step t, evaluated in double
// so that a launch sequence with constant scalar arguments (a replayed recording, replay.hip) is still a correct NEXT step.
// The registered pointer is process-wide like the precision word: an engine registers its state for the duration of its
// step (launchers read the pointer when they LAUNCH; kernels read the state when they run).
static std::atomic<const void*> g_seed_epoch{nullptr};
extern "C" int focr_set_seed_epoch(const void* state) {
  g_seed_epoch.store(state, std::memory_order_relaxed);
  return FOCR_OK;
}
extern "C" const uint64_t* focr_seed_epoch(void) {
  return reinterpret_cast<const uint64_t*>(g_seed_epoch.load(std::memory_order_relaxed));
}

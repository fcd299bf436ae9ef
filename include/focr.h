/* libfocr_hip.so -- C ABI of the MI355X (gfx950) kernels behind the FudanOCR
 * scene-text-telescope / text-gestalt SR + CRNN-CTC training hot path.
 *
 * The reference has NO native/FFI layer for this path (pure Python/PyTorch):
 * each entry point below replaces a torch op at the cited reference call site (paths under
 * /root/reference/scene-text-telescope/).  The binding a reference maintainer would add is a
 * ctypes stub; the in-tree one is fudanocr_amd/_lib.py.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer to fp32 data unless noted;
 *  - activations are channel-last: [N,H,W,C] or [rows,C], contiguous;
 *  - conv weights are [Cout][KH][KW][Cin] (a torch channels_last [Cout,Cin,KH,KW] tensor);
 *  - the library never allocates, frees or retains device memory; workspaces are caller-owned;
 *  - every call is asynchronous on `stream`, re-entrant, and never synchronises the device;
 *  - return 0 on success, <0 on failure (FOCR_E*); message via focr_last_error() (thread-local).
 */
#ifndef FOCR_H
#define FOCR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* focr_stream_t; /* hipStream_t */

#define FOCR_OK 0
#define FOCR_EINVAL (-1)
#define FOCR_EUNSUPPORTED (-2)
#define FOCR_EHIP (-3)
#define FOCR_ENCCL (-4) /* RCCL: call failed, asynchronous communicator error, or watchdog timeout */

#define FOCR_ACT_NONE 0
#define FOCR_ACT_RELU 1
#define FOCR_ACT_MISH 4

const char* focr_last_error(void);
int focr_version(void);
/* contraction precision of the kernels that have both paths (process-wide):
 *   0 = exact fp32 on the f32-input MFMA; 1 = split bf16 "bf16x3" (hi/lo operands, 3 products, fp32 accumulate) on the
 *   bf16 MFMA pipe -- same end-to-end error as fp32 (tools/exp_split_precision.py); 2 (default) = 1 with single-bf16
 *   gradient accumulations in the attention backward; 3 = 2 with single-bf16 data-gradient products: the data-gradient
 *   convolutions on the halo kernel and dP = dO V^T in the single-pass attention backward (S stays split).  Forward results are identical in modes 1-3 (csrc/focr_core.hip). */
int focr_set_precision(int mode);
/* A/B kernel-selection switches for measurements (results do not depend on them; defaults are the production
 * kernels).  key 0: transformer-linear weight gradients on the streaming kernel (1, default) or the generic split
 * kernel (0); key 1: attention forward with 256-query (1, default) / 128-query (0) blocks / look-ahead scores (2);
 * key 2: persistent LSTM scan, one launch per layer (1, default; 2 = the same with the agent-scope release in every step even
 * when a group's 8 blocks share an XCD) or one launch per time step (0);
 * key 3: attention backward: 2 = single pass, dQ / dK / dV from one S / dP evaluation (precision modes 2 / 3
 * and Ntok % 256 == 0, otherwise as 1); 1 = two passes (dK/dV, then dQ with 256-query blocks); 0 = two passes,
 * 128-query dQ blocks; 4 (default) = 2 when the launch has at least 128 (batch, head) blocks, else 1 (one block per (batch, head)
 * leaves most CUs idle at small batches); key 4: the 256-query attention forward: 2 (default) = keep-word scalar requests behind the K-fragment
 * reads (travelling under the score MFMAs) + softmax on scores relative to the running reference (cross-half max exchange
 * only in the rescale branch); 1 = the keep-word schedule alone; 0 = requests in front of the fragment reads (round 1-4).
 * Values 0 and 1 are bit-identical; 2 agrees with them to rounding whenever a rescale happens.
 * key 5: TSRN GRU scans (focr_gru_bidir_*): 2 (default) = 16-sequence compute waves on the 16x16x32 MFMA + one loader wave
 * that DMAs the next steps' operands into an LDS ring; 1 = 32-sequence compute wave + two loader waves; 0 = single-wave scans
 * with register prefetch.  0 and 1 are bit-identical, 2 agrees with them to rounding. */
int focr_set_tuning(int key, int value);
int focr_get_tuning(int key);
int focr_get_precision(void);

/* ---- convolution / linear: nn.Conv2d, nn.Linear (stride 1) --------------------------------
 * model/tsrn.py:26-43,80-94,101-114,132  model/tbsrn.py:74,103,117-129,158-163
 * model/stn_head.py:13-49,88-99          model/crnn/crnn.py:31-63,12,19
 * y = [relu](alpha * conv(x,w) + bias + residual); ldy/ldr/ldx = row pitch of y/residual/x
 * (0: dense).
 * dgrad = the same call on focr_weight_flip_transpose()d weights with pad' = K-1-pad. */
int focr_conv2d_fwd(const float* x, const float* w, const float* bias, const float* residual, float* y,
                    int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW,
                    float alpha, int relu, int ldy, int ldr, int ldx, focr_stream_t stream);
/* The same convolution with caller-provided scratch: layers with few output tiles and a long contraction (the STN
 * head's 1x2 .. 2x8 px maps, the CRNN tail) are split along K, every split writes its tile to a slot of `ws` and the
 * slots are folded in a fixed order (deterministic, no atomics).  focr_conv2d_fwd_ws_floats: scratch size in floats,
 * 0 = the layer does not split (then focr_conv2d_fwd_ws == focr_conv2d_fwd). */
long focr_conv2d_fwd_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW);
int focr_conv2d_fwd_ws(const float* x, const float* w, const float* bias, const float* residual, float* y, int N,
                       int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW, float alpha, int relu,
                       int ldy, int ldr, int ldx, float* ws, long ws_floats, focr_stream_t stream);
/* y = Dropout_p(relu(alpha * x W^T + b)) for a Linear [rows,Cin] -> [rows,Cout] (PositionwiseFeedForward,
 * tbsrn.py:162-163), dropout fused into the GEMM epilogue.  *keep_scale (host) receives 1/P(keep) (P quantised to
 * 1/65536): dropped elements are exactly the zeros of y, so the backward is focr_relu_bwd_scaled(dy, y, ., keep_scale). */
int focr_linear_relu_dropout_fwd(const float* x, const float* w, const float* bias, float* y, long rows, int Cin,
                                 int Cout, float alpha, float p_drop, uint64_t seed, float* keep_scale,
                                 focr_stream_t stream);
/* g = (mask_src > 0) ? scale * (x W^T) : 0: the data gradient of a Linear whose input h = mask_src came out of a relu
 * (or fused relu + dropout) Linear, with that producer's relu backward in the epilogue (w: [Cout][Cin], the transposed
 * weight of the layer whose data gradient this is).  Returns FOCR_EUNSUPPORTED (-2) for shapes the streaming kernel
 * does not take; the caller then runs focr_conv2d_fwd + focr_relu_bwd_scaled. */
int focr_linear_masked_fwd(const float* x, const float* w, const float* mask_src, float* y, long rows, int Cin,
                           int Cout, float scale, focr_stream_t stream);
/* dw[Cout][KH][KW][Cin], dbias[Cout] (nullable); ldd = row pitch of dy (0: Cout).  Gradient outputs of
 * every *_wgrad/_bwd entry are accumulated with atomics: prezeroed=0 clears them first (overwrite),
 * prezeroed=1 means the caller guarantees zeros (slices of a gradient buffer cleared once per step). */
/* ws (nullable) / ws_floats: focr_conv2d_wgrad_ws_floats() floats of scratch; with it the layers that have a
 * partial-tile path (3x3, Cin = 64 in the bf16x3 modes) write per-block partial tiles and fold them in a second
 * kernel instead of sending every block's tile through same-address atomics. */
long focr_conv2d_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int padH, int padW);
int focr_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W,
                      int Cin, int Cout, int KH, int KW, int padH, int padW, int ldd, int ldx,
                      int prezeroed, float* ws, long ws_floats, focr_stream_t stream);
/* w[Cout][KH][KW][Cin] -> wd[Cin][KH][KW][Cout] with both spatial axes flipped */
int focr_weight_flip_transpose(const float* w, float* wd, int Cout, int KH, int KW, int Cin,
                               focr_stream_t stream);
/* the same for n layers in one launch: descs_dev = device array of n focr_flip_desc */
typedef struct focr_flip_desc {
  const float* w; /* [Cout][KH][KW][Cin] */
  float* wd;      /* [Cin][KH][KW][Cout], taps flipped */
  int cout, kh, kw, cin;
} focr_flip_desc;
int focr_weight_flip_transpose_batched(const void* descs_dev, int n, int max_elems, focr_stream_t stream);

/* ---- 3x3 / pad 1 convolution with the input tile resident in LDS ("halo" kernel, csrc/conv3x3_halo.hip) --------
 * Same layers as focr_conv2d_fwd when KH = KW = 3, pad 1, Cin % 64 == 0, Cout % 64 == 0: SRB conv1/conv2 and block7
 * (model/tbsrn.py:232-251, model/tsrn.py:77-98), the upsample conv (model/tsrn.py:104-114), the CRNN body
 * (model/crnn/crnn.py:31-63) and their data gradients.  The weights are consumed PRE-SPLIT (bf16 hi/lo planes) and in
 * MFMA-fragment order: focr_weight_prep_frag* produce that form once per weight update --
 *   piece (n / 32, k / 16, plane) = 1 KB, lane 32 * ((k >> 3) & 1) + (n & 31) holds k & 7 (8 bf16);
 *   flip = 0: rows n = Cout, k = (kh, kw, ci)       (forward);
 *   flip = 1: rows n = Cin,  k = (KH-1-kh, KW-1-kw, co)  (data gradient: the call below on dy with Cin/Cout swapped).
 * focr_weight_frag_bytes(rows, k): size of that buffer.  planes = 2: split products (fp32-equivalent); 1: single bf16.
 * stats (nullable): float [focr_conv3x3_frag_tiles(N,H,W)][Cout][2] per-tile (sum, sum of squares) of the stored
 * outputs, for the BatchNorm that follows (model/tbsrn.py:234,237; folded by focr_bn_train_fwd_stats). */
long focr_weight_frag_bytes(int rows, int k);
int focr_weight_prep_frag(const float* w, void* wfrag, int Cout, int KH, int KW, int Cin, int flip,
                          focr_stream_t stream);
typedef struct focr_wprep_desc {
  const float* w; /* [Cout][KH][KW][Cin] */
  void* wfrag;
  int cout, kh, kw, cin, flip, pad_;
} focr_wprep_desc;
int focr_weight_prep_frag_batched(const void* descs_dev, int n, long max_threads, focr_stream_t stream);
int focr_conv3x3_frag_tiles(int N, int H, int W);
int focr_conv3x3_frag_fwd(const float* x, const void* wfrag, const float* bias, const float* residual, float* y,
                          float* stats, int N, int H, int W, int Cin, int Cout, float alpha, int relu, int planes,
                          int ldy, int ldr, int ldx, focr_stream_t stream);
/* the same launch with a relu-backward mask in its epilogue: the output (the data gradient of a layer whose INPUT was a relu
 * output) is kept where mask[pixel][channel] > 0, else 0 -- the producing layer's relu backward without its own pass
 * (ResNet basic blocks of the frozen focus-loss recognizers, loss/transformer.py:90-139); mask: [N][H][W][ldm >= Cout] fp32 */
int focr_conv3x3_frag_fwd_masked(const float* x, const void* wfrag, const float* bias, const float* residual, float* y,
                                 int N, int H, int W, int Cin, int Cout, float alpha, int planes, int ldy, int ldr, int ldx,
                                 const float* mask, int ldm, focr_stream_t stream);
/* out[c] = sum_r x[r*ld + c]   (bias gradients) */
int focr_colsum(const float* x, float* out, long rows, int C, int ld, focr_stream_t stream);
/* specialised 9x9, pad 4, Cin=64 -> Cout<=3 convolution (SR output layer, model/tsrn.py:43):
 * the 9 horizontal taps are folded into the MFMA N dimension ((co,kw) = 27 of 32 columns).  Cout == 4 (the reference's
 * --mask, main.py:31) in precision modes 1-3: two launches of two channels each, forward and the _ws weight gradient */
int focr_conv9x9_small_cout_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H,
                                int W, int Cin, int Cout, focr_stream_t stream);
int focr_conv9x9_small_cout_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H,
                                  int W, int Cin, int Cout, int prezeroed, focr_stream_t stream);
/* the same gradient on the bf16 matrix pipe with split operands (precision modes 1-3), bias gradient from the same pass;
 * dw / dbias are overwritten; ws: focr_conv9x9_small_cout_wgrad_ws_floats(N, H, W, Cout) floats (per-block partial tiles,
 * folded in a fixed order: no atomics).  model/tsrn.py:43 / tbsrn.py:197 backward. */
long focr_conv9x9_small_cout_wgrad_ws_floats(int N, int H, int W, int Cout);
int focr_conv9x9_small_cout_wgrad_ws(const float* x, const float* dy, float* dw, float* dbias, float* ws, long ws_floats,
                                     int N, int H, int W, int Cin, int Cout, focr_stream_t stream);

/* ---- fused attention: model/tbsrn.py:132-150 (+ the head split/merge of :116-126) -----------
 * q,k,v (and dq,dk,dv): row pitch ld; o, d_o: row pitch ldo (0 = ld) -- q,k,v may be column slices of one packed
 * [B,Ntok,3*128] projection; head h in columns h*32..h*32+31; lse: [B,H,Ntok]; Ntok % 128 == 0.
 * p_drop: dropout on the probabilities (tbsrn.py:147-148), quantised to 1/4096; the forward draws the mask from a counter
 * hash seeded stream into mask: uint32 [B*H, Ntok/32 query groups, Ntok/32 key groups, 32 slots]; bit j of a word =
 * query 32*qg+j, the word's key = 32*kg+kk with slot = 2*((kk&3)+4*(kk>>3)) + ((kk>>2)&1)
 * (B*H*Ntok*Ntok/32 words, needed iff p>0). */
int focr_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse,
                       uint32_t* mask, int B, int H, int Ntok, int ld, int ldo, float scale, float p_drop,
                       uint64_t seed, focr_stream_t stream);
/* The keep bits depend on (B, H, Ntok, p_drop, seed) only: focr_attention_dropout_mask draws them (on any stream, e.g.
 * ahead of time beside other work), focr_attention_fwd_premasked is the forward on bits that are already in `mask`
 * (same results as focr_attention_fwd with that seed). */
int focr_attention_dropout_mask(uint32_t* mask, int B, int H, int Ntok, float p_drop, uint64_t seed,
                                focr_stream_t stream);
int focr_attention_fwd_premasked(const float* q, const float* k, const float* v, float* o, float* lse,
                                 const uint32_t* mask, int B, int H, int Ntok, int ld, int ldo, float scale,
                                 float p_drop, focr_stream_t stream);
/* dwork: B*H*Ntok floats */
/* 1 / P(keep) of the attention dropout (P(keep) = 1 - p_drop quantised to 1/4096); 1 when p_drop rounds to "no dropout" */
float focr_attention_keep_scale(float p_drop);
/* o == NULL: dwork already holds D = rowsum(d_o * o) per (b, head, token) (see focr_fe_post_bwd) */
int focr_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                       const float* lse, const uint32_t* mask, float* dq, float* dk, float* dv,
                       float* dwork, int B, int H, int Ntok, int ld, int ldo, float scale, float p_drop,
                       focr_stream_t stream);

/* ---- BatchNorm2d/1d (+activation, +residual): model/tsrn.py:81-86,35-39, stn_head.py:17-21,45-48,
 *      crnn.py:44 ; torch semantics (biased var to normalise, unbiased in the running update) --- */
long focr_bn_ws_floats(long rows, int C);     /* workspace size (floats) of focr_bn_train_fwd */
int focr_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, long long* num_batches_tracked, const float* residual,
                      float* y, float* save_mean, float* save_invstd, float* ws, long rows, int C,
                      float momentum, float eps, int act, focr_stream_t stream);
/* train-mode forward with the batch statistics taken from per-tile partial sums written by the producing convolution
 * (focr_conv3x3_frag_fwd `stats`: part[nparts][C][2] = (sum, sum of squares)): no statistics pass over x. */
int focr_bn_train_fwd_stats(const float* x, const float* part, int nparts, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* num_batches_tracked,
                            const float* residual, float* y, float* save_mean, float* save_invstd, long rows, int C,
                            float momentum, float eps, int act, focr_stream_t stream);
int focr_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, const float* residual, float* y, float* invstd_out,
                     long rows, int C, float eps, int act, focr_stream_t stream);
/* the same with invstd = focr_bn_eval_fwd's invstd_out of an earlier call (same running_var, eps): frozen statistics need
 * the 1/sqrt(var + eps) pass once, not every step */
int focr_bn_eval_apply(const float* x, const float* gamma, const float* beta, const float* running_mean,
                       const float* invstd, const float* residual, float* y, long rows, int C, int act,
                       focr_stream_t stream);
int focr_bn_bwd(const float* dz, const float* x, const float* gamma, const float* beta, const float* mean,
                const float* invstd, float* dx, float* dgamma, float* dbeta, float* ws, long rows, int C,
                int act, int train, int lddz, focr_stream_t stream);
/* lddz: row pitch of dz in floats (0 = C): dz may be a column slice of a wider matrix (the gradient of the feature half
 * of the [feature | positional encoding] token matrix, tbsrn.py:85).
 * train = 0 (statistics were the running ones): dx = gamma * invstd * act'(.) dz; dgamma / dbeta / ws may be null
 * (frozen layer) or all given (trainable affine parameters under eval statistics). */
long focr_bn_bwd_ws_floats(long rows, int C); /* workspace size (floats) of focr_bn_bwd */

/* ---- the reference's own LayerNorm (unbiased std, eps on std): model/tbsrn.py:23-36 ---------- */
int focr_layernorm_fwd(const float* x, const float* residual, const float* a, const float* b, float* y,
                       float* save_mean, float* save_rinv, long rows, int D, float eps,
                       focr_stream_t stream);
int focr_layernorm_bwd(const float* dy, const float* x, const float* residual, const float* a,
                       const float* save_mean, const float* save_rinv, float* dx, float* da, float* db,
                       long rows, int D, float eps, int prezeroed, focr_stream_t stream);

/* ---- activations / layout -------------------------------------------------------------------
 * PReLU (single slope) tsrn.py:28; PixelShuffle(2)+mish tsrn.py:101-125; tanh tsrn.py:73 */
int focr_prelu_fwd(const float* x, const float* slope, float* y, long n, focr_stream_t stream);
int focr_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* dslope, long n,
                   int prezeroed, focr_stream_t stream);
int focr_pixelshuffle_mish_fwd(const float* pre, float* z, int N, int H, int W, int C, focr_stream_t stream);
int focr_pixelshuffle_mish_bwd(const float* dz, const float* pre, float* dpre, int N, int H, int W, int C,
                               focr_stream_t stream);
int focr_nchw_to_nhwc(const float* x, float* y, int N, int C, int HW, focr_stream_t stream);
int focr_nhwc_to_nchw(const float* x, float* y, int N, int C, int HW, int do_tanh, focr_stream_t stream);
int focr_tanh_bwd_to_nhwc(const float* dy_nchw, const float* y_nchw, float* dx_nhwc, int N, int C, int HW,
                          focr_stream_t stream);
int focr_relu_bwd(const float* dy, const float* y, float* dx, long n, focr_stream_t stream);
int focr_relu_bwd_scaled(const float* dy, const float* y, float* dx, long n, float scale, focr_stream_t stream);
/* tokens = [feat | positional encoding] tbsrn.py:83-86 ; column slice (+add) for its backward */
int focr_concat_pe(const float* feat, const float* pe, float* tok, long rows, int Cf, int Cp, int T,
                   focr_stream_t stream);
int focr_slice_cols(const float* x, const float* add, float* out, long rows, int ld, int c0, int w,
                    focr_stream_t stream);
/* ---- fused row-local chains of the FeatureEnhancer (model/tbsrn.py:76-92: everything between the attention output
 * and the block's 128 -> 64 projection is independent per token row; csrc/fe_chain.hip).  rows = B * H * W tokens
 * (multiple of 32), d_model 128, precision mode != 0.  Kept for the backward: the NORMALISED rows xhat1 / xhat2 and
 * 1 / (std + eps) of the two LayerNorms (tbsrn.py:23-36) and h = Dropout(relu(w_1 .)).
 *   focr_fe_post_fwd : [ctx -> O-proj (+ tok) -> LN1 -> w_1 -> relu -> dropout] [-> w_2 (+ LN1 out) -> LN3 -> linear
 *                      128 -> 64 (+ xin)], two launches; *keep_scale (host) receives 1 / P(keep) of the FFN dropout.
 *   focr_fe_post_bwd : the two data-gradient chains (d_out -> d_s2, d_hpre; -> d_s1 = gradient of LN1's input sum = the
 *                      O-proj output gradient = the token's residual gradient, d_ctx = attention output gradient).
 *                      With dwork != NULL it also writes D[b][head][token] = sum_d d_ctx * ctx (ntok tokens per image),
 *                      the row term of the attention backward: focr_attention_bwd is then called with o = NULL.
 *   focr_fe_qkv_fwd  : tok = [feat | pe[row % ntok]] (tbsrn.py:83-86) and the packed q | k | v projection in one kernel.
 *                      planes != NULL (optional; NULL in the product): Q * q_mul, K, V are (also, or with qkv = NULL:
 *                      only) written pre-split: [rows][3 x 256] bf16, every four columns of a [rows][128] fp32 tensor as
 *                      [hi x 4 | lo x 4] (x = hi + lo, hi = bf16(x): the bytes of the fp32 row) -- the operand form of
 *                      the PL attention kernels, an experiment kept in tools/ubench; focr_fe_post_bwd's d_ctx_planes
 *                      likewise receives d_ctx * planes_mul as [rows][256] (d_ctx itself may then be NULL).
 *   focr_fe_qkv_dgrad: d_feat[rows,64] = dqkv[rows,384] Wqkv[:, 0:64] + d_s1[:, 0:64] (the positional-encoding half of
 *                      the token, tbsrn.py:83-86, has no gradient consumer).
 *   focr_fe_wgrads   : every parameter gradient of these layers in one call (targets are overwritten); the LayerNorm
 *                      a_2 / b_2 gradients and the weight gradients of the linears fed by a LayerNorm come out of one
 *                      weight-gradient GEMM on xhat.  ws: focr_fe_wgrads_ws_floats(rows) floats.  parts: bit 0 = the
 *                      four linears fed by the backward chains (may run before the attention backward), bit 1 = the
 *                      packed q | k | v projection (needs dqkv); operands of an unselected part may be NULL. */
int focr_fe_chain_supported(long rows, int d_model);
int focr_fe_post_fwd(const float* ctx, const float* tok, const float* xin, const float* wo, const float* bo,
                     const float* a1, const float* b1, const float* w1, const float* bb1, const float* w2,
                     const float* bb2, const float* a3, const float* b3, const float* wl, const float* bl,
                     float* xhat1, float* rinv1, float* h, float* xhat2, float* rinv2, float* out, long rows,
                     float eps, float p_drop, uint64_t seed, float* keep_scale, focr_stream_t stream);
int focr_fe_post_bwd(const float* d_out, const float* wl, const float* xhat2, const float* rinv2, const float* a3,
                     const float* w2, const float* h, float keep_scale, const float* w1, const float* xhat1,
                     const float* rinv1, const float* a1, const float* wo, float* d_s2, float* d_hpre, float* d_s1,
                     float* d_ctx, long rows, float eps, const float* ctx, float* dwork, int ntok,
                     void* d_ctx_planes, float planes_mul, focr_stream_t stream);
int focr_fe_qkv_fwd(const float* feat, const float* pe, const float* wqkv, const float* bqkv, float* tok,
                    float* qkv, long rows, int ntok, void* planes, float q_mul, focr_stream_t stream);
/* the same with `feat` = the INPUT of the train-mode BatchNorm2d(64) in front of the block's FeatureEnhancer (tbsrn.py:246-251):
 * gamma (x - mean) invstd + beta is applied on load (statistics from focr_bn_train_fwd_stats with y = NULL); the normalised
 * tensor is never written */
int focr_fe_qkv_fwd_bn(const float* feat, const float* pe, const float* wqkv, const float* bqkv, float* tok,
                       float* qkv, long rows, int ntok, void* planes, float q_mul, const float* bn_gamma,
                       const float* bn_beta, const float* bn_mean, const float* bn_invstd, focr_stream_t stream);
int focr_fe_qkv_dgrad(const float* dqkv, const float* wqkv, const float* d_s1, float* d_feat, long rows,
                      focr_stream_t stream);
long focr_fe_wgrads_ws_floats(long rows);
int focr_fe_wgrads(const float* d_out, const float* xhat2, const float* d_s2, const float* h, const float* d_hpre,
                   const float* xhat1, const float* d_s1, const float* ctx, const float* dqkv, const float* tok,
                   const float* wl, const float* w1, const float* a1, const float* b1, const float* a3,
                   const float* b3, float* g_wl, float* g_bl, float* g_a3, float* g_b3, float* g_w2, float* g_bb2,
                   float* g_w1, float* g_bb1, float* g_a1, float* g_b1, float* g_wo, float* g_bo, float* g_wqkv,
                   float* g_bqkv, float* ws, long ws_floats, long rows, int parts, focr_stream_t stream);
/* nn.Dropout tbsrn.py:160,163 : y = keep ? x/(1-p) : 0 ; the same call is its backward */
int focr_dropout(const float* x, float* y, long n, float p, uint64_t seed, focr_stream_t stream);
/* nn.MSELoss loss/text_focus_loss.py:44,86 ; upstream = device scalar */
int focr_mse_fwd(const float* a, const float* b, float* out, long n, focr_stream_t stream);
int focr_mse_bwd(const float* a, const float* b, const float* upstream, float* da, long n,
                 focr_stream_t stream);
int focr_axpy(const float* x, const float* add, float* y, long n, float alpha, focr_stream_t stream);
int focr_scale_dev(const float* x, const float* s, float* y, long n, focr_stream_t stream);

/* ---- pooling / resampling -------------------------------------------------------------------
 * nn.MaxPool2d stn_head.py:34-42, crnn.py:52-62 (idx: uint8 window-local argmax) */
int focr_maxpool_fwd(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, int kh, int kw,
                     int sh, int sw, int ph, int pw, focr_stream_t stream);
int focr_maxpool_bwd(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, int kh,
                     int kw, int sh, int sw, int ph, int pw, focr_stream_t stream);
/* the same, fused with the backward of the relu in front of the pooling layer (model/crnn/crnn.py:52-63 conv -> relu -> pool):
 * ypool = the pooling forward output; gradients of windows whose maximum is 0 are dropped (relu'(y <= 0) = 0) */
int focr_maxpool_relu_bwd(const float* dy, const uint8_t* idx, const float* ypool, float* dx, int N, int H, int W, int C,
                          int kh, int kw, int sh, int sw, int ph, int pw, focr_stream_t stream);
/* First recognizer layer fused with its activation and pooling: Conv2d(1, 64, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2)
 * (model/crnn/crnn.py:51-52 convRelu(0) + pooling0), frozen-recognizer form (interfaces/super_resolution.py:168-171: the
 * training step needs d loss / d input only).  x [N, H, W, 1]; w [64][3][3][1] = nn.Conv2d's [64, 1, 3, 3]; bias [64] or
 * NULL; y, ypool, dy [N, H/2, W/2, 64]; idx uint8 window-local argmax (focr_maxpool_fwd's rule); dx [N, H, W, 1].
 * H % 4 == 0, W even and <= 128.  fp32 arithmetic in every precision mode. */
int focr_crnn_conv0_pool_supported(int H, int W, int Cin, int Cout, int KH, int KW, int pad);
int focr_crnn_conv0_pool_fwd(const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int N, int H,
                             int W, focr_stream_t stream);
int focr_crnn_conv0_pool_bwd(const float* dy, const uint8_t* idx, const float* ypool, const float* w, float* dx, int N,
                             int H, int W, focr_stream_t stream);
/* TPS grid + F.grid_sample: model/tps_spatial_transformer.py:97-111,10-18 ; src: [B,H*W,2] */
int focr_tps_fwd(const float* img, const float* ctrl, const float* inv_kernel, const float* coord_repr,
                 float* out, float* src, int B, int H, int W, int C, int NC, focr_stream_t stream);
int focr_tps_bwd(const float* dout, const float* img, const float* src, const float* inv_kernel,
                 const float* coord_repr, float* dctrl, int B, int H, int W, int C, int NC,
                 focr_stream_t stream);
/* gradient of the same warp w.r.t. the sampled image (F.grid_sample backward w.r.t. input,
 * model/tps_spatial_transformer.py:10-18): dimg[B,H,W,C] is overwritten; src = the sampling coordinates focr_tps_fwd
 * saved. */
int focr_tps_bwd_img(const float* dout, const float* src, float* dimg, int B, int H, int W, int C,
                     focr_stream_t stream);
/* parse_crnn_data: interfaces/base.py:319-325 ; x NCHW [B,Cx>=3,H,IW] -> y [B,H,OW] */
int focr_bicubic_gray_fwd(const float* x_nchw, float* y, int B, int Cx, int H, int IW, int OW,
                          focr_stream_t stream);
int focr_bicubic_gray_bwd(const float* dy, float* dx_nchw, int B, int Cx, int H, int IW, int OW,
                          focr_stream_t stream);

/* ---- recurrences ---------------------------------------------------------------------------
 * nn.LSTM(bidirectional) crnn.py:11,15 : gx [rows][2][4H] (row(t,b) = t*st_t + b*st_b),
 * whh [2][4H][H], bhh [2][4H], hseq [T][B][2H], gates [T][B][2][4H], cseq [T][B][2][H] */
long focr_lstm_ws_bytes(int T, int B, int H, int backward);  /* bf16 operand-copy workspace (bf16x3 mode) */
int focr_lstm_bidir_fwd(const float* gx, const float* whh, const float* bhh, float* hseq, float* gates,
                        float* cseq, void* ws, int T, int B, int H, int st_t, int st_b, focr_stream_t stream);
int focr_lstm_bidir_bwd(const float* dhseq, const float* whh, const float* gates, const float* cseq,
                        float* dgx, float* dc_carry /*2*B*H*/, void* ws, int T, int B, int H, int st_t,
                        int st_b, focr_stream_t stream);
/* The same scans with the hi / lo split of W_hh prepared ONCE (focr_lstm_prepare_weights: backward = 0 for the forward scan's
 * [2][4H][H] form, 1 for the backward scan's transposed form; focr_lstm_split_bytes(H) bytes each): a caller whose recurrent
 * weights do not change between calls -- the frozen recognizer of the training step, super_resolution.py:168-171 -- skips
 * the per-call split launch.  pflags (optional, FOCR_LSTM_FLAG_BYTES bytes, zeroed ONCE by the caller and then owned by one
 * (weights, B, T) call site) + base: the persistent scans' step counters keep counting across calls instead of being cleared
 * by a memset launch in front of every scan; every call advances each group's word by 8 (T - 1) and the caller passes the sum
 * so far (mod 2^32) as `base`.  wsplit == NULL and pflags == NULL: identical to the entries above. */
#define FOCR_LSTM_FLAG_BYTES 1024
int focr_lstm_persistent_usable(int B, int H);  /* 1: this shape runs as one persistent launch (pflags / base are used) */
long focr_lstm_split_bytes(int H);
int focr_lstm_prepare_weights(const float* whh, void* out, int H, int backward, focr_stream_t stream);
int focr_lstm_bidir_fwd_pw(const float* gx, const float* whh, const float* bhh, float* hseq, float* gates, float* cseq,
                           void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H, int st_t,
                           int st_b, focr_stream_t stream);
int focr_lstm_bidir_bwd_pw(const float* dhseq, const float* whh, const float* gates, const float* cseq, float* dgx,
                           float* dc_carry, void* ws, const void* wsplit, void* pflags, unsigned base, int T, int B, int H,
                           int st_t, int st_b, focr_stream_t stream);
/* nn.GRU(64, 32, bidirectional, batch_first) tsrn.py:133,141 (gate order r,z,n).  All tensors are
 * indexed by map row: row(seq n, time t) = (n/IC)*OS + (n%IC)*IS + t*TS, so both the horizontal
 * (gru2) and the vertical (gru1, reference transposes the map) scans read the NHWC map in place.
 * gx [rows][2][96] (= x W_ih^T + b_ih), whh [2][96][32], bhh [2][96], hseq [rows][64],
 * gates [rows][2][128] = (r,z,n, W_hn h + b_hn).  Backward writes dgx (grad of gx), dgh (grad of
 * W_hh h + b_hh: rows r,z as dgx, row n = d hn) and hprev [rows][2][32] (h_{t-1}) for the W_hh wgrad. */
int focr_gru_bidir_fwd(const float* gx, const float* whh, const float* bhh, float* hseq, float* gates,
                       int nseq, int T, int IC, int OS, int IS, int TS, focr_stream_t stream);
int focr_gru_bidir_bwd(const float* dhseq, const float* whh, const float* gates, const float* hseq,
                       float* dgx, float* dgh, float* hprev, int nseq, int T, int IC, int OS, int IS,
                       int TS, focr_stream_t stream);
/* dW_hh [2][96][32] of both directions = the diagonal blocks of cross [192][64] = dgh^T [h_prev(dir 0) | h_prev(dir 1)], the
 * weight gradient ONE focr_conv2d_wgrad call (1x1, Cin = 64, Cout = 192, dbias = cross + 192 * 64) produces from the two
 * tensors focr_gru_bidir_bwd wrote; cross[192 * 64 ..] = that call's 192 bias sums = db_hh [2][96] (dbhh may be NULL);
 * accumulate != 0 adds to dwhh / dbhh. */
int focr_gru_whh_extract(const float* cross, float* dwhh, float* dbhh, int accumulate, focr_stream_t stream);

/* ---- CTC: log_softmax + F.ctc_loss(blank 0, 'mean', zero_infinity) (label codec
 *      utils/utils_crnn.py:21-53).  logits/grad [T,B,C]; loss: 1 float; nll: B floats ------------ */
int focr_ctc_fwd(const float* logits, const int* targets, const int* target_lengths,
                 const int* target_offsets, float* loss, float* nll, float* grad_logits, int T, int B,
                 int C, focr_stream_t stream);

/* ---- optimiser tail: clip_grad_norm_(0.25) + Adam (interfaces/super_resolution.py:83-84,
 *      interfaces/base.py:194-198) on flat buffers; gscale = 1/world for DP averaging ----------- */
long focr_grad_sumsq_ws_floats(void);   /* size of the sumsq workspace; sumsq[0] = squared norm (no atomics:
                                          bit-identical on every data-parallel rank) */
int focr_grad_sumsq(const float* g, float* sumsq, long n, float gscale, focr_stream_t stream);
int focr_clip_adam(float* p, const float* g, float* m, float* v, const float* sumsq, long n, float lr,
                   float beta1, float beta2, float eps, int step, float max_norm, float gscale,
                   focr_stream_t stream);
/* ---- device-resident step state: 64 bytes of caller-owned, zero-initialised device memory holding what changes from one
 * optimisation step to the next (interfaces/super_resolution.py:79-84 runs these as host-side Python state: nn.Dropout's
 * generator, Adam's `state['step']`): u64 dropout epoch, i64 optimiser step count t, f32 1 - beta1^t, f32 sqrt(1 - beta2^t)
 * (evaluated in double).  focr_step_advance: epoch + 1, t + 1, corrections of the new t -- the first launch of a step.
 * focr_clip_adam_state: focr_clip_adam with t / the corrections read from the state.  focr_set_seed_epoch: process-wide
 * registration (NULL to clear): every launch that takes a dropout seed (focr_attention_dropout_mask, focr_fe_post_fwd,
 * focr_linear_relu_dropout_fwd, focr_dropout, focr_small_attention_fwd) hands the pointer to its kernel, which folds the
 * epoch word into the seed ON THE DEVICE.  With both, a step's launches carry the same scalar arguments in every step. */
int focr_step_state_bytes(void);
int focr_step_advance(void* state, double beta1, double beta2, focr_stream_t stream);
int focr_clip_adam_state(float* p, const float* g, float* m, float* v, const float* sumsq, long n, float lr, float beta1,
                         float beta2, float eps, const void* state, float max_norm, float gscale, focr_stream_t stream);
int focr_set_seed_epoch(const void* state);
/* optimizer.zero_grad() on the flat gradient buffer (interfaces/super_resolution.py:82); p 16-byte aligned, n floats */
int focr_zero(float* p, long n, focr_stream_t stream);

/* ---- evaluation metrics on the device (utils/ssim_psnr.py:9-78; interfaces/super_resolution.py:178-181) ------
 * One pass over an NCHW pair (first 3 channels): sq_sum[b] = sum (255 a - 255 b)^2, ssim_sum[b] = sum of the SSIM map
 * (window_size-tap normalised Gaussian `window_host`, a HOST array; zero padding; C1 = 0.01^2, C2 = 0.03^2).
 * ws: focr_psnr_ssim_ws_floats(B, H, W) floats. */
long focr_psnr_ssim_ws_floats(int B, int H, int W);
int focr_psnr_ssim(const float* img1, const float* img2, const float* window_host, int window_size, float* sq_sum,
                   float* ssim_sum, float* ws, int B, int C, int H, int W, focr_stream_t stream);

/* ---- input pipeline, device half of resizeNormalize (dataset/dataset.py:136-152) ------------------------------
 * uint8 [B,H,W,3] -> float32 [B,3(+1),H,W]: ToTensor (/255) and, with mask = 1, the mean-threshold mask channel
 * (PIL integer luma, 1.0 where luma <= the image's mean luma).  Bit-identical to the reference's CPU transform. */
int focr_u8_to_input(const unsigned char* u8_nhwc, float* out_nchw, int B, int H, int W, int mask,
                     focr_stream_t stream);

/* ---- stroke-level-decomposition transformer recognizer (BASELINE configs[4]) --------------------------------
 * /root/reference/stroke-level-decomposition/model/transformer.py, train.py.  The encoder's 3x3 convolutions, BatchNorm,
 * max-pool, linears, LayerNorm (any width) and dropout reuse the entry points above; these are the additional ops. */
/* y = relu(a + b): BasicBlock tail, transformer.py:66-75 (backward: focr_relu_bwd(dy, y, .) for both addends) */
int focr_add_relu_fwd(const float* a, const float* b, float* y, long n, focr_stream_t stream);
/* y[r] = table[idx[r]] * scale (Embeddings, transformer.py:269-277; idx int64); bwd ACCUMULATES into dtable */
int focr_embedding_fwd(const long long* idx, const float* table, float* y, long rows, int D, float scale,
                       focr_stream_t stream);
int focr_embedding_bwd(const long long* idx, const float* dy, float* dtable, long rows, int D, float scale,
                       focr_stream_t stream);
/* ragged gather of prediction rows (transformer.py:362-370): scatter = 0: dst[r] = src[idx[r]]; 1: dst[idx[r]] = src[r] */
int focr_gather_rows(const float* src, const long long* idx, float* dst, long rows, int D, int scatter,
                     focr_stream_t stream);
/* nn.CrossEntropyLoss (train.py:41,70): loss[0] = mean nll, grad = d loss / d logits; nll_ws: rows floats */
int focr_cross_entropy_fwd(const float* logits, const long long* target, float* loss, float* nll_ws, float* grad,
                           long rows, int C, focr_stream_t stream);
/* torch.optim.Adadelta(lr, rho, eps) fused over flat buffers (train.py:36-38); gscale = 1/world */
int focr_adadelta(float* p, const float* g, float* sq, float* acc, long n, float lr, float rho, float eps,
                  float gscale, focr_stream_t stream);
/* attention with few queries and 64- or 256-wide heads (MultiHeadedAttention/attention, transformer.py:184-238): q [B,Lq,H*Dk],
 * k/v [B,Lk,H*Dk] (row pitches ldq/ldk, o: ldo), optional causal mask, dropout on the probabilities.  p / pd
 * [B,H,Lq,Lk]: softmax before / after dropout (pd is the reference's returned attention map); ws: B*H*Lq*Lk floats. */
int focr_small_attention_fwd(const float* q, const float* k, const float* v, float* o, float* p, float* pd, int B, int H,
                             int Lq, int Lk, int Dk, int ldq, int ldk, int ldo, float scale, int causal, float p_drop,
                             uint64_t seed, focr_stream_t stream);
int focr_small_attention_bwd(const float* q, const float* k, const float* v, const float* d_o, const float* p,
                             const float* pd, const float* dmap, float* dq, float* dk, float* dv, float* ws, int B,
                             int H, int Lq, int Lk, int Dk, int ldq, int ldk, int ldo, float scale, int causal,
                             focr_stream_t stream);
/* ---- text-focus loss (scene-text-telescope/loss/text_focus_loss.py:84-99, loss/weight_ce_loss.py:38-45) ------------
 * the frozen recognizer (loss/transformer.py) reuses the convolution / BatchNorm / attention (Dk = 64, dmap = the
 * gradient arriving through the returned attention map) / LayerNorm entries; these are the criterion's own ops:
 * mean |a - b| (ws: 256 floats) and its gradient w.r.t. b; weighted cross-entropy with a [C][C] weight table. */
int focr_l1_fwd(const float* a, const float* b, float* out, float* ws, long n, focr_stream_t stream);
int focr_l1_bwd(const float* a, const float* b, const float* g, float* db, long n, focr_stream_t stream);
int focr_weight_cross_entropy_fwd(const float* logits, const long long* target, const float* table, float* loss,
                                  float* nll_ws, float* grad, long rows, int C, focr_stream_t stream);
/* the same two losses on PADDED label layouts (recordable step: shapes do not depend on the batch's labels).  plan: device
 * int64 [>= 2] = {longest label of the batch, number of real label positions}.  L1: a, b [outer][L][inner], positions
 * j >= plan[0] are left out of the sum, the mean (outer * plan[0] * inner entries) and the gradient; weighted CE: rows with
 * target < 0 are padding, mean over plan[1] rows.  text_focus_loss.py:62-99 with L = a capacity instead of max(len). */
int focr_l1_masked_fwd(const float* a, const float* b, float* out, float* ws, long outer, int L, int inner,
                       const long long* plan, focr_stream_t stream);
int focr_l1_masked_bwd(const float* a, const float* b, const float* g, float* db, long outer, int L, int inner,
                       const long long* plan, focr_stream_t stream);
int focr_weight_cross_entropy_masked_fwd(const float* logits, const long long* target, const float* table, float* loss,
                                         float* nll_ws, float* grad, long rows, int C, const long long* plan,
                                         focr_stream_t stream);

/* ---- data-parallel gradient exchange (RCCL over xGMI; replaces nn.DataParallel's gather of the gradients,
 * reference interfaces/base.py:178-179).  One communicator per process = per GPU.  RCCL is bound at run time.
 * focr_comm_unique_id: rank 0 draws the 128-byte id (sizeof(ncclUniqueId)) and gives it to every rank by a host channel;
 * focr_comm_init: collective, on the thread whose current device is the rank's GPU;
 * focr_allreduce_async: buf[0..n) <- sum over ranks, in place, on `stream` (dtype 0 = fp32);
 * focr_comm_nranks: 0 when there is no communicator;
 * focr_comm_count: the rank count RCCL reports for the live communicator (ncclCommCount; 0 without one);
 * focr_comm_rccl_version: ncclGetVersion's code of the RCCL instance the library is bound to (negative: cannot load);
 * focr_comm_async_error: non-blocking ncclCommGetAsyncError (also checked in front of every focr_allreduce_async);
 * focr_comm_wait: host-side watchdog -- waits for `stream` to drain while polling the communicator; on an asynchronous
 *   error or after timeout_ms (<= 0: no limit) the communicator is aborted (ncclCommAbort) and FOCR_ENCCL returned. */
#define FOCR_COMM_ID_BYTES 128
int focr_comm_unique_id(void* id_out);
int focr_comm_init(int rank, int nranks, const void* unique_id);
int focr_allreduce_async(void* buf, size_t n, int dtype, focr_stream_t stream);
int focr_comm_nranks(void);
int focr_comm_count(void);
int focr_comm_rccl_version(void);
int focr_comm_async_error(void);
int focr_comm_wait(focr_stream_t stream, int timeout_ms);
int focr_comm_destroy(void);

/* ---- recorded step (csrc/replay.hip): the launches of one whole optimisation step -- TextSR.train's loop body,
 * interfaces/super_resolution.py:66-84 -- captured once into a hipGraph_t by the caller (HIP stream capture; the graph is
 * never instantiated) and re-issued from a loop inside the library: ONE host call per step.
 * focr_replay_build: graph = hipGraph_t; lanes = n_lanes hipStream_t the captured chains are laid out on (longest chain on
 *   lanes[0]); kernel / memset / linear-memcpy / empty nodes only, else FOCR_EUNSUPPORTED.  The graph (it owns the kernel
 *   argument storage) must outlive the handle; the handle owns HIP events only.
 * focr_replay_launch: re-issue everything, ordered on `stream` (NULL: lanes[0]) like one launch on that stream.
 * focr_replay_info: out[8] = nodes, kernel / memset / memcpy / empty nodes, lanes used, cross-lane waits, events.
 * focr_replay_lanes / focr_replay_node_name: per-node lane and (mangled) kernel name, for tools.
 * focr_replay_probe: `depth` timing event pairs around every kernel node whose name contains `pattern` (launch k records
 *   pair k % depth); returns the number of nodes matched (>= 0).  focr_replay_probe_read: per probe the MEAN milliseconds
 *   over the launches its pairs hold (caller has synchronised), the node index and that launch count; returns the number
 *   of probes. */
int focr_replay_build(void* graph, void* const* lanes, int n_lanes, void** handle);
int focr_replay_launch(void* handle, focr_stream_t stream);
int focr_replay_info(void* handle, int* out);
int focr_replay_lanes(void* handle, int* lane, int n);
int focr_replay_node_name(void* handle, int i, char* buf, int n);
int focr_replay_probe(void* handle, const char* pattern, int depth);
int focr_replay_probe_read(void* handle, float* ms, int* node, int* count, int n);
int focr_replay_destroy(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* FOCR_H */

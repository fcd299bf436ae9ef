"""ctypes loader for the C-ABI library libfocr_hip.so (include/focr.h).

There is NO fallback: if the library is missing or a symbol is absent, importing the ops
raises.  The product never routes through the CPU oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FOCR_LIB overrides the library path (kernel A/B experiments with tools/kbench.py only)
LIB_PATH = os.environ.get("FOCR_LIB") or os.path.join(_HERE, "libfocr_hip.so")

P, I, L, F, U = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64
U32 = ctypes.c_uint32

# name -> argument types (every entry point returns int and takes the stream last)
SIGNATURES = {
    "focr_conv2d_fwd": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, I, P],
    "focr_conv2d_fwd_ws": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, I, I, I, I, P, L, P],
    "focr_conv2d_fwd_ws_floats": [I, I, I, I, I, I, I, I, I],
    "focr_linear_masked_fwd": [P, P, P, P, L, I, I, F, P],
    "focr_conv2d_wgrad": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, P, L, P],
    "focr_conv2d_wgrad_ws_floats": [I, I, I, I, I, I, I, I, I],
    "focr_weight_flip_transpose": [P, P, I, I, I, I, P],
    "focr_weight_flip_transpose_batched": [P, I, I, P],
    "focr_colsum": [P, P, L, I, I, P],
    "focr_weight_frag_bytes": [I, I],
    "focr_weight_prep_frag": [P, P, I, I, I, I, I, P],
    "focr_weight_prep_frag_batched": [P, I, L, P],
    "focr_conv3x3_frag_tiles": [I, I, I],
    "focr_conv3x3_frag_fwd": [P, P, P, P, P, P, I, I, I, I, I, F, I, I, I, I, I, P],
    "focr_conv3x3_frag_fwd_masked": [P, P, P, P, P, I, I, I, I, I, F, I, I, I, I, P, I, P],
    "focr_attention_fwd": [P, P, P, P, P, P, I, I, I, I, I, F, F, U, P],
    "focr_attention_dropout_mask": [P, I, I, I, F, U, P],
    "focr_attention_fwd_premasked": [P, P, P, P, P, P, I, I, I, I, I, F, F, P],
    "focr_attention_bwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, F, P],
    "focr_bn_train_fwd": [P, P, P, P, P, P, P, P, P, P, P, L, I, F, F, I, P],
    "focr_bn_train_fwd_stats": [P, P, I, P, P, P, P, P, P, P, P, P, L, I, F, F, I, P],
    "focr_bn_eval_fwd": [P, P, P, P, P, P, P, P, L, I, F, I, P],
    "focr_bn_eval_apply": [P, P, P, P, P, P, P, L, I, I, P],
    "focr_bn_bwd": [P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P],
    "focr_layernorm_fwd": [P, P, P, P, P, P, P, L, I, F, P],
    "focr_layernorm_bwd": [P, P, P, P, P, P, P, P, P, L, I, F, I, P],
    "focr_prelu_fwd": [P, P, P, L, P],
    "focr_prelu_bwd": [P, P, P, P, P, L, I, P],
    "focr_pixelshuffle_mish_fwd": [P, P, I, I, I, I, P],
    "focr_pixelshuffle_mish_bwd": [P, P, P, I, I, I, I, P],
    "focr_nchw_to_nhwc": [P, P, I, I, I, P],
    "focr_nhwc_to_nchw": [P, P, I, I, I, I, P],
    "focr_tanh_bwd_to_nhwc": [P, P, P, I, I, I, P],
    "focr_concat_pe": [P, P, P, L, I, I, I, P],
    "focr_slice_cols": [P, P, P, L, I, I, I, P],
    "focr_dropout": [P, P, L, F, U, P],
    "focr_fe_chain_supported": [L, I],
    "focr_fe_post_fwd": [P] * 21 + [L, F, F, U, P, P],
    "focr_fe_post_bwd": [P] * 7 + [F] + [P] * 9 + [L, F, P, P, I, P, F, P],
    "focr_fe_qkv_fwd": [P, P, P, P, P, P, L, I, P, F, P],
    "focr_fe_qkv_fwd_bn": [P, P, P, P, P, P, L, I, P, F, P, P, P, P, P],
    "focr_attention_keep_scale": [F],
    "focr_fe_qkv_dgrad": [P, P, P, P, L, P],
    "focr_fe_wgrads_ws_floats": [L],
    "focr_fe_wgrads": [P] * 31 + [L, L, I, P],
    "focr_mse_fwd": [P, P, P, L, P],
    "focr_mse_bwd": [P, P, P, P, L, P],
    "focr_axpy": [P, P, P, L, F, P],
    "focr_relu_bwd": [P, P, P, L, P],
    "focr_relu_bwd_scaled": [P, P, P, L, F, P],
    "focr_linear_relu_dropout_fwd": [P, P, P, P, L, I, I, F, F, U, P, P],
    "focr_maxpool_fwd": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "focr_maxpool_bwd": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "focr_maxpool_relu_bwd": [P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "focr_crnn_conv0_pool_supported": [I, I, I, I, I, I, I],
    "focr_crnn_conv0_pool_fwd": [P, P, P, P, P, I, I, I, P],
    "focr_crnn_conv0_pool_bwd": [P, P, P, P, P, I, I, I, P],
    "focr_tps_fwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "focr_tps_bwd": [P, P, P, P, P, P, I, I, I, I, I, P],
    "focr_tps_bwd_img": [P, P, P, I, I, I, I, P],
    "focr_bicubic_gray_fwd": [P, P, I, I, I, I, I, P],
    "focr_bicubic_gray_bwd": [P, P, I, I, I, I, I, P],
    "focr_lstm_bidir_fwd": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "focr_lstm_bidir_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "focr_lstm_bidir_fwd_pw": [P, P, P, P, P, P, P, P, P, U32, I, I, I, I, I, P],
    "focr_lstm_bidir_bwd_pw": [P, P, P, P, P, P, P, P, P, U32, I, I, I, I, I, P],
    "focr_lstm_prepare_weights": [P, P, I, I, P],
    "focr_lstm_split_bytes": [I],
    "focr_lstm_persistent_usable": [I, I],
    "focr_gru_bidir_fwd": [P, P, P, P, P, I, I, I, I, I, I, P],
    "focr_gru_bidir_bwd": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "focr_gru_whh_extract": [P, P, P, I, P],
    "focr_conv9x9_small_cout_fwd": [P, P, P, P, I, I, I, I, I, P],
    "focr_conv9x9_small_cout_wgrad": [P, P, P, P, I, I, I, I, I, I, P],
    "focr_conv9x9_small_cout_wgrad_ws": [P, P, P, P, P, L, I, I, I, I, I, P],
    "focr_conv9x9_small_cout_wgrad_ws_floats": [I, I, I, I],
    "focr_ctc_fwd": [P, P, P, P, P, P, P, I, I, I, P],
    "focr_scale_dev": [P, P, P, L, P],
    "focr_grad_sumsq": [P, P, L, F, P],
    "focr_bn_ws_floats": [L, I],
    "focr_bn_bwd_ws_floats": [L, I],
    "focr_lstm_ws_bytes": [I, I, I, I],
    "focr_grad_sumsq_ws_floats": [],
    "focr_psnr_ssim_ws_floats": [I, I, I],
    "focr_psnr_ssim": [P, P, P, I, P, P, P, I, I, I, I, P],
    "focr_u8_to_input": [P, P, I, I, I, I, P],
    "focr_add_relu_fwd": [P, P, P, L, P],
    "focr_embedding_fwd": [P, P, P, L, I, F, P],
    "focr_embedding_bwd": [P, P, P, L, I, F, P],
    "focr_gather_rows": [P, P, P, L, I, I, P],
    "focr_cross_entropy_fwd": [P, P, P, P, P, L, I, P],
    "focr_adadelta": [P, P, P, P, L, F, F, F, F, P],
    "focr_small_attention_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, F, U, P],
    "focr_small_attention_bwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, F, I, P],
    "focr_l1_fwd": [P, P, P, P, L, P],
    "focr_l1_bwd": [P, P, P, P, L, P],
    "focr_weight_cross_entropy_fwd": [P, P, P, P, P, P, L, I, P],
    "focr_l1_masked_fwd": [P, P, P, P, L, I, I, P, P],
    "focr_l1_masked_bwd": [P, P, P, P, L, I, I, P, P],
    "focr_weight_cross_entropy_masked_fwd": [P, P, P, P, P, P, L, I, P, P],
    "focr_set_precision": [I],
    "focr_set_tuning": [I, I],
    "focr_comm_unique_id": [P],
    "focr_comm_init": [I, I, P],
    "focr_allreduce_async": [P, L, I, P],
    "focr_comm_nranks": [],
    "focr_comm_count": [],
    "focr_comm_rccl_version": [],
    "focr_comm_async_error": [],
    "focr_comm_wait": [P, I],
    "focr_comm_destroy": [],
    "focr_get_tuning": [I],
    "focr_get_precision": [],
    "focr_clip_adam": [P, P, P, P, P, L, F, F, F, F, I, F, F, P],
    "focr_zero": [P, L, P],
    "focr_step_state_bytes": [],
    "focr_step_advance": [P, ctypes.c_double, ctypes.c_double, P],
    "focr_clip_adam_state": [P, P, P, P, P, L, F, F, F, F, P, F, F, P],
    "focr_set_seed_epoch": [P],
    "focr_replay_build": [P, P, I, P],
    "focr_replay_info": [P, P],
    "focr_replay_lanes": [P, P, I],
    "focr_replay_launch": [P, P],
    "focr_replay_node_name": [P, I, P, I],
    "focr_replay_probe": [P, ctypes.c_char_p, I],
    "focr_replay_probe_read": [P, P, P, P, I],
    "focr_replay_destroy": [P],
}

_lib = None


def load():
    """Load the library and bind every symbol declared in include/focr.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fudanocr_amd: %s is missing -- build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64 and the library must bind to THAT runtime (the one that owns torch's
    # streams and allocations).  Loading libfocr_hip.so before torch pulls /opt/rocm's copy in beside it and every
    # launch then fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = ctypes.c_int
        _FN[name] = fn
    lib.focr_last_error.restype = ctypes.c_char_p
    lib.focr_last_error.argtypes = []
    lib.focr_version.restype = ctypes.c_int
    lib.focr_version.argtypes = []
    lib.focr_bn_ws_floats.restype = ctypes.c_long
    lib.focr_bn_bwd_ws_floats.restype = ctypes.c_long
    lib.focr_lstm_ws_bytes.restype = ctypes.c_long
    lib.focr_lstm_split_bytes.restype = ctypes.c_long
    lib.focr_grad_sumsq_ws_floats.restype = ctypes.c_long
    lib.focr_conv2d_wgrad_ws_floats.restype = ctypes.c_long
    lib.focr_conv9x9_small_cout_wgrad_ws_floats.restype = ctypes.c_long
    lib.focr_conv2d_fwd_ws_floats.restype = ctypes.c_long
    lib.focr_weight_frag_bytes.restype = ctypes.c_long
    lib.focr_psnr_ssim_ws_floats.restype = ctypes.c_long
    lib.focr_fe_wgrads_ws_floats.restype = ctypes.c_long
    lib.focr_attention_keep_scale.restype = ctypes.c_float
    _lib = lib
    if os.environ.get("FOCR_PRECISION"):          # 0 fp32 | 1 bf16x3 | 2 (default) + bf16 attention-gradient sums | 3 + bf16 dgrad
        rc = lib.focr_set_precision(int(os.environ["FOCR_PRECISION"]))
        if rc != 0:
            raise RuntimeError("FOCR_PRECISION: " + lib.focr_last_error().decode())
    return lib


# optional on-stream timing of selected entry points (bench.py's roofline leg): name -> list of
# (start_event, end_event) recorded on the stream the kernel is launched on.
_timed = None


def start_timing(names):
    global _timed
    _timed = {n: [] for n in names}


def add_timing(names):
    """start timing more entry points while a timing session is open"""
    for n in names:
        _timed.setdefault(n, [])


def stop_timing():
    """Returns {name: [ms, ...]} (synchronises); with_args=True -> {name: [(ms, call args), ...]}."""
    return _stop_timing(False)


def stop_timing_with_args():
    return _stop_timing(True)


def _stop_timing(with_args):
    global _timed
    import torch
    torch.cuda.synchronize()
    if with_args:
        out = {n: [(a.elapsed_time(b), args) for a, b, args in ev] for n, ev in (_timed or {}).items()}
    else:
        out = {n: [a.elapsed_time(b) for a, b, _ in ev] for n, ev in (_timed or {}).items()}
    _timed = None
    return out


def set_precision(mode):
    """0 = fp32 MFMA, 1 = split-bf16 ("bf16x3") MFMA, 2 = 1 + single-bf16 gradient accumulations in the attention
    backward, 3 = 2 + single-bf16 data-gradient convolutions (csrc/focr_core.hip)."""
    call("focr_set_precision", int(mode))


def get_precision():
    return load().focr_get_precision()


_FN = {}          # name -> bound ctypes function (filled by load(): no attribute lookup per call)


def call(name, *args):
    if _timed is None and _lib is not None:          # the hot path: 360 calls per training step
        rc = _FN[name](*args)
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (name, rc, _lib.focr_last_error().decode()))
        return
    lib = load()
    if _timed is not None and name in _timed:
        import torch
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib, name)(*args)
        b.record()
        _timed[name].append((a, b, args))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.focr_last_error().decode()))

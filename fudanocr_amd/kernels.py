"""torch.autograd.Function wrappers over the C-ABI HIP library (include/focr.h).

PyTorch is plumbing here: it owns device memory (caching allocator), the stream and the
autograd tape; every arithmetic op of the path runs in a hand-written gfx950 kernel.
Internal activation layout is channel-last: [N,H,W,C] / [rows,C], fp32, contiguous.
Convolution weights are `[Cout,Cin,KH,KW]` tensors held in channels_last memory format,
i.e. physically [Cout][KH][KW][Cin] -- exactly the K-contiguous operand the implicit-GEMM
kernel wants, with the reference's state_dict shapes unchanged.
"""
import contextlib
import ctypes
import os
import math
import threading
import weakref

import numpy as np
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_MISH = 0, 1, 4
_NULL = None          # ctypes converts None -> NULL and a Python int -> the pointer for `c_void_p` argtypes: the helpers below hand
#                       out plain ints (no ctypes object per argument: ~600 of them per step on a host-bound path)


def _p(t):
    return None if t is None else t.data_ptr()


def _po(t, off):
    """device pointer `off` floats into tensor t"""
    return t.data_ptr() + 4 * off


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """the current HIP stream of the current device as a C pointer.  (torch.cuda.current_stream() builds a Stream object
    through three Python layers: ~9 us, 270 times per step; the raw getter is one C call.)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("fudanocr_amd kernel needs a contiguous fp32 CUDA tensor, got %s %s %s"
                               % (t.device, t.dtype, tuple(t.stride())))


def _ohwi(w):
    """[Cout,Cin,KH,KW] parameter -> physically-OHWI flat storage (no copy when the parameter
    is already channels_last)."""
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _target(param):
    """Flat-gradient slice the engine attached to a parameter (engine.FlatBuffers), or None.
    When present the backward kernels write the parameter gradient straight into it and hand
    autograd `None` (no per-parameter accumulate kernels; every parameter is used once per step)."""
    return getattr(param, "_focr_grad", None) if param is not None else None


def _new_seed(ctx=None):
    """dropout seed of the next dropout site: the owning StepContext's deterministic per-site sequence when it has a
    `seed_base` (engine steps: the per-STEP variation then comes from the device-resident epoch word, focr_set_seed_epoch),
    a host random number otherwise"""
    return (ctx or current_context()).next_seed()


def _mix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


# ----------------------------------------------------------------------------------------
# convolution / linear
# ----------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------
# Deferred residual gradients.  When a tensor t feeds BOTH a conv/linear (as its input) and a residual slot of a
# later op, autograd would add the two gradients of t with a separate elementwise kernel (three 67 MB passes).  The
# residual consumer always runs its backward first (it is downstream of the GEMM), so it can park its gradient here
# and return None; the GEMM's data-gradient kernel then adds it in its epilogue (`residual` operand) and returns the
# complete gradient.  The model code opts in on both sides (defer=True / take_deferred=True) -- it knows the wiring.
# A parked gradient that is never picked up would be a silent error: check_deferred() raises at the next forward.
# ----------------------------------------------------------------------------------------
class StepContext:
    """Per-owner (engine instance) state of the autograd wrappers: parked residual gradients, the flipped / fragment
    weight tables and the weight-gradient side stream.  Forward passes pick the context up from a THREAD-LOCAL slot
    (`use_context`), and every autograd node remembers the context it was recorded under, so its backward -- which
    autograd may run on a worker thread -- finds the same state: two engines, or concurrent backward threads, in
    one process never share a table."""

    def __init__(self, managed_frags=False):
        self.deferred = {}
        self.premasked = {}              # data_ptr of gradients whose relu backward a consumer's dgrad already applied
        self.flips = None                # FlipTable while an engine step's backward runs
        self.frags = None                # managed FragTable while an engine step runs (None: the default table)
        self.side_enabled = False
        self.side_stream_obj = None
        self.side_used = False
        self.mask_prefetch = False       # engine-owned contexts draw the attention keep bits ahead of time
        self._masks = []                 # per attention call of a step: {"key", "mask", "event"}
        self._mask_cursor = 0
        self._early_done = False
        self._pending_side = []          # weight-gradient launches parked for a better moment (defer_side)
        self._tail_side = []             # ... parked until the backward reaches the STN head (park_tail)
        self._fe_fwd_n = 0               # FeatureEnhancer blocks seen by this step's forward / backward
        self._fe_bwd_i = 0
        self._parking = False
        self.tail_ready = False          # the step has a TPS warp with a trainable STN head behind it (release point)
        self.seed_base = None            # engine-owned contexts: site seeds are a fixed sequence, the step's epoch lives on
        self._seed_i = 0                 # the device (replayed recordings carry constant scalar arguments)

    def next_seed(self):
        if self.seed_base is None:
            return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())
        self._seed_i += 1
        return _mix64(self.seed_base + self._seed_i) >> 2

    # -- attention dropout keep bits.  They depend on (shape, p, seed) only, so the engine draws the bits of ALL
    # attention layers at the start of the step on the side stream (idle during the forward pass) instead of in front
    # of every attention forward on the main stream (5 x 58 us at the bench shape).
    def next_mask(self, b, heads, t, p, device):
        """keep-bit tensor of the next attention call of this step -> (mask, ready)"""
        i = self._mask_cursor
        self._mask_cursor += 1
        key = (b, heads, t, float(p), device)
        if i < len(self._masks) and self._masks[i]["key"] == key:
            m = self._masks[i]
            if m["event"] is not None:
                torch.cuda.current_stream().wait_event(m["event"])
                m["event"] = None
                return m["mask"], True
            return m["mask"], False
        mask = torch.empty((b, heads, t // 32, t // 32, 32), device=device, dtype=torch.int32)
        if self.mask_prefetch:
            del self._masks[i:]
            self._masks.append({"key": key, "mask": mask, "event": None})
        return mask, False

    def prefetch_masks(self):
        """start of an engine step: every recorded mask either was drawn EARLY into its second buffer (see
        `prefetch_masks_early`: the buffers are swapped) or is redrawn now on the side stream.  The buffers redrawn
        here are the ones the previous step's backward read; that backward is already queued on the current stream,
        which the side stream waits for first."""
        self._mask_cursor = 0
        self._early_done = False
        self.new_step()
        if not (self.mask_prefetch and self._masks):
            return
        late = [m for m in self._masks if m.get("alt_event") is None]
        for m in self._masks:
            if m.get("alt_event") is not None:
                m["mask"], m["alt"] = m["alt"], m["mask"]
                m["event"], m["alt_event"] = m["alt_event"], None
        if late:
            self._draw_masks(late, "mask", "event")

    def _draw_masks(self, entries, buf, evt):
        if self.side_stream_obj is None:
            self.side_stream_obj = torch.cuda.Stream()
        s = self.side_stream_obj
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for m in entries:
                b, heads, t, p, _ = m["key"]
                _lib.call("focr_attention_dropout_mask", _p(m[buf]), b, heads, t, p, _new_seed(self), _stream())
                ev = torch.cuda.Event()
                ev.record(s)
                m[evt] = ev

    def prefetch_masks_early(self):
        """called in front of the recognizer's LSTM scan (a latency-bound persistent kernel that leaves HBM and most CUs
        idle for ~0.4 ms): the keep bits of the NEXT step are drawn now, on the side stream, into a second buffer per
        attention call -- the current step's backward still reads the first one.  The second buffer was last read by
        the backward of the step before this one, which the current stream (that the side stream waits for) has behind
        it.  A step without such a call (SR-only configurations) draws at the step start as before."""
        if not (self.mask_prefetch and _MASK_EARLY and self._masks) or self._early_done:
            return
        if self._mask_cursor != len(self._masks):          # not every attention call of this step has happened yet
            return
        self._early_done = True
        for m in self._masks:
            if m.get("alt") is None:
                m["alt"] = torch.empty_like(m["mask"])
        self._draw_masks(self._masks, "alt", "alt_event")

    # -- deferred residual gradients
    def defer_grad(self, t, g):
        k = _dkey(t)
        if k in self.deferred:
            raise RuntimeError("two deferred gradients for the same tensor")
        self.deferred[k] = g

    def take_deferred(self, t):
        return self.deferred.pop(_dkey(t), None)

    def mark_premasked(self, t):
        """`t` is a gradient whose producer already applied the consumer layer's relu backward.  Keyed by address but
        matched by storage identity + shape (a weak reference): an entry nobody consumed -- an autograd.grad on an
        intermediate outside the engine -- can never be taken for a later tensor that reuses the address."""
        self.premasked[t.data_ptr()] = (weakref.ref(t.untyped_storage()), t.numel())

    def take_premasked(self, t):
        hit = self.premasked.pop(t.data_ptr(), None)
        if hit is None:
            return False
        st = hit[0]()
        return st is not None and st is t.untyped_storage() and hit[1] == t.numel()

    def check_deferred(self):
        self.premasked.clear()
        if self.deferred:
            self.deferred.clear()
            raise RuntimeError("fudanocr_amd: a deferred residual gradient was never consumed by its GEMM backward")

    # -- weight-gradient side stream
    def side_stream(self):
        if not self.side_enabled:
            return None
        if self.side_stream_obj is None:
            self.side_stream_obj = torch.cuda.Stream()
        self.side_used = True
        return self.side_stream_obj

    # -- parked side-stream work (experiment switch FOCR_DEFER_SIDE, off by default -- see _DEFER_SIDE): a residual block
    # may park the weight gradients of its convolutions and of its QKV projection, and the NEXT block (in backward order)
    # issues them when its attention backward starts.  Every joiner of the side stream flushes first, so nothing can be
    # left behind.
    def defer_side(self, fn):
        if self.side_enabled and _DEFER_SIDE:
            self._pending_side.append(fn)
        else:
            fn()

    def flush_side(self):
        while self._pending_side:
            self._pending_side.pop(0)()

    # -- weight gradients parked for the TAIL of the backward pass (FOCR_PARK_TAIL = k, see _PARK_TAIL): the STN head's
    # backward is ~60 dependent launches of 4 - 20 us that leave the chip almost idle; the weight gradients of the k
    # residual blocks that come last in backward order are issued when the backward reaches the TPS warp instead of beside
    # those blocks' own (HBM-bound, LDS-full) main-stream kernels.  Released by _TPSWarp.backward, by every joiner of the
    # side stream and at the end of the engine's backward: nothing can be left behind.
    def fe_backward_begins(self):
        self._fe_bwd_i += 1
        self._parking = bool(self.side_enabled and self.tail_ready and _PARK_TAIL > 0 and
                             self._fe_bwd_i > self._fe_fwd_n - _PARK_TAIL)

    def park_tail(self, fn, otherwise=None):
        """park `fn` for the tail if this block is one of the last k, else hand it to `otherwise` (default: run now)"""
        if self._parking and self.side_enabled:
            self._tail_side.append(fn)
        elif otherwise is not None:
            otherwise(fn)
        else:
            fn()

    def flush_tail(self):
        self._parking = False
        while self._tail_side:
            self._tail_side.pop(0)()

    def new_step(self):
        self._seed_i = 0
        self._fe_fwd_n = self._fe_bwd_i = 0
        self._parking = False
        self.tail_ready = False

    def join_side_stream(self, stream=None):
        """make `stream` (default: current) wait for everything queued on the weight-gradient side stream"""
        self.flush_side()
        self.flush_tail()
        if self.side_used and self.side_stream_obj is not None:
            (stream or torch.cuda.current_stream()).wait_stream(self.side_stream_obj)


_DEFAULT_CTX = StepContext()
_TLS = threading.local()


def current_context():
    return getattr(_TLS, "ctx", None) or _DEFAULT_CTX


@contextlib.contextmanager
def use_context(c):
    prev = getattr(_TLS, "ctx", None)
    _TLS.ctx = c
    try:
        yield c
    finally:
        _TLS.ctx = prev


def _dkey(t):
    return (t.data_ptr(), t.numel())


def check_deferred():
    """raise if a deferred residual gradient of the current context was never consumed (called at the start of a
    model forward and by the engine after backward)"""
    current_context().check_deferred()


# ----------------------------------------------------------------------------------------
# Flipped / transposed weights for the data-gradient GEMMs.  Weights do not change during backward, so the training
# engine flips every layer in ONE launch per step (FlipTable) instead of one ~6 us launch in front of each dgrad;
# frozen weights (the recogniser) are flipped once.  The first step registers the layers as their backward runs.
# ----------------------------------------------------------------------------------------
class FlipTable:
    def __init__(self):
        self.reg = {}            # weight data_ptr -> (cout, kh, kw, cin, frozen)
        self.views = {}          # weight data_ptr -> flipped weights (valid for the current step)
        self.frozen_ver = {}     # frozen weight data_ptr -> tensor version its cached flip belongs to
        self.built = False
        self.n_live = 0

    def lookup(self, wk, cout, kh, kw, cin, frozen):
        ptr = wk.data_ptr()
        wd = self.views.get(ptr)
        if wd is None:
            if not self.built:
                self.reg[ptr] = (cout, kh, kw, cin, bool(frozen))
                if frozen:
                    self.frozen_ver[ptr] = wk._version
            return None
        if frozen and self.frozen_ver.get(ptr) != wk._version:     # frozen weights were overwritten (e.g. a checkpoint
            return None                                            # was loaded): fall back to a fresh per-layer flip
        return wd

    def _descs(self, items, device):
        import numpy as np
        dt = np.dtype([("w", "<u8"), ("wd", "<u8"), ("cout", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4")])
        arr = np.zeros(len(items), dtype=dt)
        for i, (ptr, view, (cout, kh, kw, cin, _)) in enumerate(items):
            arr[i] = (ptr, view.data_ptr(), cout, kh, kw, cin)
        return torch.from_numpy(arr.view(np.uint8).copy()).to(device)

    def build(self, device):
        """after the first backward: one buffer for all flipped weights + the device descriptor tables"""
        if self.built or not self.reg:
            return
        total = sum(c * a * b * d for c, a, b, d, _ in self.reg.values())
        self.buf = torch.empty(total, device=device, dtype=torch.float32)
        off, live, frozen = 0, [], []
        for ptr, geo in self.reg.items():
            n = geo[0] * geo[1] * geo[2] * geo[3]
            view = self.buf[off:off + n]
            off += n
            (frozen if geo[4] else live).append((ptr, view, geo))
            self.views[ptr] = view
        self.max_live = max([g[0] * g[1] * g[2] * g[3] for _, _, g in live], default=0)
        self.n_live = len(live)
        self.live_desc = self._descs(live, device) if live else None
        if frozen:                                   # frozen layers: flipped once, here
            fd = self._descs(frozen, device)
            _lib.call("focr_weight_flip_transpose_batched", _p(fd), len(frozen),
                      max(g[0] * g[1] * g[2] * g[3] for _, _, g in frozen), _stream())
            self._keep = fd
        self.built = True

    def refresh(self):
        """once per step, before backward: re-flip the trainable layers"""
        if self.built and self.n_live:
            _lib.call("focr_weight_flip_transpose_batched", _p(self.live_desc), self.n_live, self.max_live, _stream())




# ----------------------------------------------------------------------------------------
# Pre-split, MFMA-fragment-ordered bf16 weights for the halo convolution kernel (csrc/conv3x3_halo.hip): the kernel
# DMAs 1 KB weight pieces straight into LDS, so the fp32 -> bf16 hi/lo split and the fragment permutation are done
# ONCE per weight update here instead of in every block of every launch.
#   * outside a training engine every entry is validated against the tensor's identity and version counter (plus the
#     engine's WEIGHT_EPOCH: the fused Adam kernel updates parameters through raw pointers, which autograd's version
#     counter does not see);
#   * inside engine.TrainStep the table is `managed`: the trainable layers registered during the first step are
#     re-prepared by ONE batched launch at the start of every later step (forward and flipped/data-gradient forms).
# Entries are keyed by the parameter OBJECT (weak reference), never by a bare data pointer, and only leaf tensors are
# cached: a per-forward temporary (torch.cat of GRU weights, a non-channels_last copy) is prepared on the spot.
# ----------------------------------------------------------------------------------------
WEIGHT_EPOCH = 0                 # bumped by the engine whenever it rewrites parameters behind autograd's back


def bump_weight_epoch():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


class _FragEntry:
    __slots__ = ("ref", "buf", "geom", "ptr", "version", "epoch", "managed", "trainable")


class FragTable:
    def __init__(self, managed=False):
        self.managed = bool(managed)
        self.entries = {}            # (id(weight), flip) -> _FragEntry
        self.live = []               # managed entries, refreshed by refresh()
        self.desc = None
        self.max_threads = 0

    @staticmethod
    def _prep(e, wk, flip):
        cout, kh, kw, cin = e.geom
        _lib.call("focr_weight_prep_frag", _p(wk), ctypes.c_void_p(e.buf.data_ptr()), cout, kh, kw, cin, int(flip),
                  _stream())

    def get(self, weight, wk, cout, kh, kw, cin, flip):
        """fragment-ordered weights of `weight` (physically OHWI tensor `wk`); flip: data-gradient form"""
        cacheable = weight.is_leaf and wk.data_ptr() == weight.data_ptr()
        key = (id(weight), bool(flip))
        e = self.entries.get(key) if cacheable else None
        geom = (cout, kh, kw, cin)
        if e is not None and (e.ref() is not weight or e.geom != geom or e.ptr != wk.data_ptr()):
            if e.managed:
                self.live = [(x, f) for x, f in self.live if x is not e]
                self.desc = None
            e = None
        if e is not None:
            if e.managed:
                return e.buf                                    # refreshed at the start of this step
            # Trainable weights are also checked against the engine's epoch (raw-pointer Adam updates are invisible to
            # ._version).  A weight that was trainable when its fragments were made and is frozen now (or the reverse)
            # is re-prepared once: a model trained by an engine and THEN frozen in-process (demo(), the reference's
            # eval) must not keep pre-training fragments.  Weights frozen all along are never touched by an engine.
            if (e.version == weight._version and e.trainable == weight.requires_grad
                    and (not weight.requires_grad or e.epoch == WEIGHT_EPOCH)):
                return e.buf
            e.version, e.epoch, e.trainable = weight._version, WEIGHT_EPOCH, weight.requires_grad
            self._prep(e, wk, flip)
            return e.buf
        e = _FragEntry()
        rows, k = (cin, kh * kw * cout) if flip else (cout, kh * kw * cin)
        e.buf = torch.empty(_lib.load().focr_weight_frag_bytes(rows, k), device=wk.device, dtype=torch.uint8)
        e.geom, e.ptr, e.version, e.epoch = geom, wk.data_ptr(), weight._version, WEIGHT_EPOCH
        e.trainable = weight.requires_grad
        e.ref = weakref.ref(weight, lambda _r, key=key: self.entries.pop(key, None)) if cacheable else (lambda: None)
        e.managed = bool(self.managed and cacheable and weight.requires_grad)
        self._prep(e, wk, flip)
        if cacheable:
            self.entries[key] = e
            if e.managed:
                self.live.append((e, bool(flip)))
                self.desc = None
        return e.buf

    def refresh(self):
        """managed tables, once per step before the forward: re-prepare every trainable layer in one launch"""
        live = [(e, f) for e, f in self.live if e.ref() is not None]
        if len(live) != len(self.live):
            self.live, self.desc = live, None
        if not self.live:
            return
        if self.desc is None:
            import numpy as np
            dt = np.dtype([("w", "<u8"), ("wf", "<u8"), ("cout", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("cin", "<i4"),
                           ("flip", "<i4"), ("pad", "<i4")])
            arr = np.zeros(len(self.live), dtype=dt)
            self.max_threads = 0
            for i, (e, f) in enumerate(self.live):
                cout, kh, kw, cin = e.geom
                arr[i] = (e.ptr, e.buf.data_ptr(), cout, kh, kw, cin, int(f), 0)
                self.max_threads = max(self.max_threads, e.buf.numel() // 32)
            self.desc = torch.from_numpy(arr.view(np.uint8).copy()).to(self.live[0][0].buf.device)
        _lib.call("focr_weight_prep_frag_batched", _p(self.desc), len(self.live), self.max_threads, _stream())


_FRAGS_DEFAULT = FragTable(managed=False)


def _frag_weights(step, weight, wk, cout, kh, kw, cin, flip):
    return (step.frags or _FRAGS_DEFAULT).get(weight, wk, cout, kh, kw, cin, flip)


def _halo_ok(h, w, cin, cout, kh, kw, ph, pw):
    """3x3 / pad 1 layers that run on the halo kernel: channel counts in multiples of 64 and a map that fills at least
    half of its 4 x 32 pixel tiles (the STN head's 2 x 8 maps stay on the generic kernel)."""
    if not (kh == 3 and kw == 3 and ph == 1 and pw == 1 and cin % 64 == 0 and cout % 64 == 0):
        return False
    if _lib.get_precision() == 0:
        return False
    return 2 * h * w >= ((h + 3) // 4 * 4) * ((w + 31) // 32 * 32)


def _is_out_layer(cin, cout, kh, kw, ph, pw, w, residual, alpha, relu):
    """SR output layer shape handled by csrc/conv9x9_out.hip (taps folded into the MFMA N dim); the four-channel layer of
    the reference's --mask (main.py:31) on the split-bf16 kernels only (two launches of two channels each)."""
    return (kh == 9 and kw == 9 and ph == 4 and pw == 4 and cin == 64 and w % 32 == 0
            and (cout <= 3 or (cout == 4 and _lib.get_precision() != 0 and _C9_WGRAD_BX3))
            and w <= 128 and residual is None and alpha == 1.0 and not relu)


def _conv_fwd_raw(x4, w_ohwi, bias, residual, cout, kh, kw, ph, pw, alpha, relu, frag=None, planes=2, stats=None,
                  mask=None):
    """frag: fragment-ordered bf16 weights (FragTable) -> halo kernel with `planes` products' worth of operand planes;
    stats: optional [tiles, cout, 2] per-tile (sum, sum of squares) output for a following BatchNorm;
    mask (halo kernel only): contiguous [n, oh, ow, cout] tensor, the output is kept where mask > 0 (a producer's relu
    backward in this data gradient's epilogue)."""
    n, h, w, cin = x4.shape
    oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
    y = torch.empty((n, oh, ow, cout), device=x4.device, dtype=torch.float32)
    if frag is not None and mask is not None:
        _lib.call("focr_conv3x3_frag_fwd_masked", _p(x4), ctypes.c_void_p(frag.data_ptr()), _p(bias), _p(residual), _p(y),
                  n, h, w, cin, cout, float(alpha), int(planes), 0, 0, 0, _p(mask), 0, _stream())
        return y
    if frag is not None:
        _lib.call("focr_conv3x3_frag_fwd", _p(x4), ctypes.c_void_p(frag.data_ptr()), _p(bias), _p(residual), _p(y),
                  _p(stats), n, h, w, cin, cout, float(alpha), int(relu), int(planes), 0, 0, 0, _stream())
        return y
    if _is_out_layer(cin, cout, kh, kw, ph, pw, w, residual, alpha, relu):
        _lib.call("focr_conv9x9_small_cout_fwd", _p(x4), _p(w_ohwi), _p(bias), _p(y), n, h, w, cin, cout,
                  _stream())
        return y
    nws = _lib.load().focr_conv2d_fwd_ws_floats(n, h, w, cin, cout, kh, kw, ph, pw)
    if nws > 0:       # few output tiles, long contraction (STN head, CRNN tail): split along K, fixed-order fold
        ws = torch.empty(nws, device=x4.device, dtype=torch.float32)
        _lib.call("focr_conv2d_fwd_ws", _p(x4), _p(w_ohwi), _p(bias), _p(residual), _p(y), n, h, w, cin, cout,
                  kh, kw, ph, pw, float(alpha), int(relu), 0, 0, 0, _p(ws), nws, _stream())
        return y
    _lib.call("focr_conv2d_fwd", _p(x4), _p(w_ohwi), _p(bias), _p(residual), _p(y), n, h, w, cin, cout,
              kh, kw, ph, pw, float(alpha), int(relu), 0, 0, 0, _stream())
    return y


class _Conv2d(torch.autograd.Function):
    """y = [relu](alpha * conv(x, w) + bias [+ residual]); x,y NHWC.  Reference: nn.Conv2d /
    nn.Linear call sites listed in csrc/conv_igemm.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, pad, alpha, relu, drop_p=0.0, take_deferred=False,
                defer_residual=False, want_stats=False, in_relu_scale=0.0):
        """want_stats (halo-kernel layers only): also returns the per-tile (sum, sum of squares) partials of y
        [tiles, Cout, 2] for a following train-mode BatchNorm (non-differentiable second output)."""
        cout = weight.shape[0]
        ctx.step = step = current_context()
        ctx.take_deferred, ctx.defer_residual = bool(take_deferred), bool(defer_residual)
        ctx.res_key = _dkey(residual) if (residual is not None and defer_residual) else None
        # x is the output h of a relu / relu-dropout Linear (scale = its 1 / P(keep)): this layer's data gradient can
        # apply that producer's relu backward in its epilogue (see backward)
        ctx.in_relu_scale = float(in_relu_scale)
        if weight.dim() == 2:
            kh = kw = 1
            ph = pw = 0
        else:
            kh, kw = weight.shape[2], weight.shape[3]
            ph, pw = pad
        ctx.x_shape = tuple(x.shape)
        lead = x.shape[:-1]
        x4 = x if x.dim() == 4 else x.reshape(-1, 1, 1, x.shape[-1])
        res4 = None if residual is None else residual.reshape(-1, cout)
        wk = _ohwi(weight)
        _chk(x4, wk, bias, res4)
        ctx.drop_scale = 0.0
        if drop_p > 0.0:
            # Dropout(relu(linear)) with the dropout fused into the GEMM epilogue (FFN, tbsrn.py:162-163)
            if not (relu and residual is None and kh == 1 and kw == 1):
                raise RuntimeError("fused dropout is only defined for relu(linear(x)) without residual")
            rows = x4.shape[0] * x4.shape[1] * x4.shape[2]
            y = torch.empty((x4.shape[0], x4.shape[1], x4.shape[2], cout), device=x4.device, dtype=torch.float32)
            ks = ctypes.c_float(0.0)
            _lib.call("focr_linear_relu_dropout_fwd", _p(x4), _p(wk), _p(bias), _p(y), rows, x4.shape[3], cout,
                      float(alpha), float(drop_p), _new_seed(), ctypes.byref(ks), _stream())
            ctx.drop_scale = float(ks.value)
        else:
            frag = stats = None
            if _halo_ok(x4.shape[1], x4.shape[2], x4.shape[3], cout, kh, kw, ph, pw):
                frag = _frag_weights(step, weight, wk, cout, kh, kw, x4.shape[3], False)
                if want_stats:
                    tiles = _lib.load().focr_conv3x3_frag_tiles(x4.shape[0], x4.shape[1], x4.shape[2])
                    stats = torch.empty((tiles, cout, 2), device=x4.device, dtype=torch.float32)
            elif want_stats:
                raise RuntimeError("want_stats needs a halo-kernel layer (conv_bn checks eligibility)")
            y = _conv_fwd_raw(x4, wk, bias, res4, cout, kh, kw, ph, pw, alpha, relu, frag=frag, stats=stats)
        ctx.geom = (kh, kw, ph, pw, float(alpha), bool(relu), bias is not None, residual is not None)
        ctx.targets = (_target(weight), _target(bias))
        ctx.save_for_backward(x4, weight, y if relu else None)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y if x.dim() == 4 else y.reshape(*lead, cout)

    @staticmethod
    def backward(ctx, dy, *_unused):
        x4, weight, y = ctx.saved_tensors
        step = ctx.step
        kh, kw, ph, pw, alpha, relu, has_bias, has_res = ctx.geom
        n, h, w, cin = x4.shape
        cout = weight.shape[0]
        oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
        dy4 = dy.contiguous().reshape(n, oh, ow, cout)
        if relu and step.take_premasked(dy4):
            pass        # the consumer's data-gradient kernel already applied (y > 0) * scale in its epilogue
        elif relu:
            g = torch.empty_like(dy4)
            if ctx.drop_scale:        # relu + fused dropout: the dropped elements are the zeros of y
                _lib.call("focr_relu_bwd_scaled", _p(dy4), _p(y), _p(g), dy4.numel(), ctx.drop_scale, _stream())
            else:
                _lib.call("focr_relu_bwd", _p(dy4), _p(y), _p(g), dy4.numel(), _stream())
            dy4 = g
        dres = dy4.reshape(dy.shape) if has_res else None
        residual_shares_dy = has_res and not (ctx.defer_residual and ctx.needs_input_grad[3])
        if has_res and ctx.defer_residual and ctx.needs_input_grad[3]:
            if ctx.res_key in step.deferred:
                raise RuntimeError("two deferred gradients for the same tensor")
            step.deferred[ctx.res_key] = dy4      # picked up by the data-gradient kernel of the tensor's other consumer
            dres = None
        wk = _ohwi(weight)
        dx = dw = db = None
        tw, tb = ctx.targets
        if ctx.needs_input_grad[1]:
            if tw is not None:
                dw = tw
            elif weight.dim() == 2:
                dw = torch.empty_like(weight, memory_format=torch.contiguous_format)
            else:   # logical [Cout,Cin,KH,KW] view of the physically-OHWI buffer the kernel fills
                dw = torch.empty((cout, kh, kw, cin), device=dy.device).permute(0, 3, 1, 2)
            need_db = has_bias and ctx.needs_input_grad[2]
            if need_db:
                db = tb if tb is not None else torch.empty(cout, device=dy.device, dtype=torch.float32)
            # targets handed out by the engine are slices of the flat gradient buffer, zeroed once per step
            pz = int(tw is not None and (db is None or tb is not None))
            # Weight gradients that land in the engine's flat buffer feed nothing inside backward: they run on the
            # side stream, concurrently with the data-gradient chain on the main stream (both kinds of kernels are
            # latency/occupancy bound, not throughput bound).  The engine joins the streams before the optimiser.
            # (a layer WITHOUT a data gradient -- the first layer of a network, i.e. the last node of backward -- has nothing on
            # the main stream to overlap with: its weight gradient runs in line, no cross-stream hand-off at the tail of the step)
            side = step.side_stream() if (tw is not None and (db is None or tb is not None) and ctx.needs_input_grad[0]) else None
            if side is not None and residual_shares_dy:
                # the residual's gradient IS dy4's storage and leaves this backward: autograd may accumulate into it in
                # place on the main stream while the side-stream weight-gradient kernel still reads it -> hand out a copy
                dres = dres.clone()
            if side is not None:
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
                x4.record_stream(side)
                dy4.record_stream(side)
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                if _is_out_layer(cin, cout, kh, kw, ph, pw, w, None, alpha, relu) and not has_res:
                    if _lib.get_precision() != 0 and w % 32 == 0 and _C9_WGRAD_BX3:
                        # split-bf16 kernel, bias gradient from the same pass (428 + 130 us -> one ~60 us launch + fold)
                        nws = _lib.load().focr_conv9x9_small_cout_wgrad_ws_floats(n, h, w, cout)
                        ws = torch.empty(nws, device=dy.device, dtype=torch.float32)
                        _lib.call("focr_conv9x9_small_cout_wgrad_ws", _p(x4), _p(dy4), _p(dw), _p(db), _p(ws), nws, n, h,
                                  w, cin, cout, _stream())
                    else:
                        _lib.call("focr_conv9x9_small_cout_wgrad", _p(x4), _p(dy4), _p(dw), _p(db), n, h, w, cin,
                                  cout, pz, _stream())
                else:
                    nws = _lib.load().focr_conv2d_wgrad_ws_floats(n, h, w, cin, cout, kh, kw, ph, pw)
                    ws = torch.empty(nws, device=dy.device, dtype=torch.float32) if nws > 0 else None
                    _lib.call("focr_conv2d_wgrad", _p(x4), _p(dy4), _p(dw), _p(db), n, h, w, cin, cout, kh, kw,
                              ph, pw, 0, 0, pz, _p(ws), nws, _stream())
                if alpha != 1.0:
                    _lib.call("focr_axpy", _p(dw), _NULL, _p(dw), dw.numel(), alpha, _stream())
        if ctx.needs_input_grad[0]:
            radd = step.take_deferred(x4) if ctx.take_deferred else None  # parked residual gradient of x: + in the epilogue
            if radd is not None:
                radd = radd.reshape(-1, cin)
            fused_mask = False
            if (ctx.in_relu_scale != 0.0 and radd is None and kh == 1 and kw == 1 and alpha == 1.0
                    and weight.dim() == 2 and x4.is_contiguous()):
                # dx = (x > 0) ? scale * (dy W) : 0 in ONE pass: the relu(-dropout) backward of the layer that produced
                # x rides in this data gradient's epilogue; that layer's backward finds dx in step.premasked and skips
                # its own relu pass (no intermediate dy_x is written or re-read)
                persistent = weight.is_leaf and wk.data_ptr() == weight.data_ptr()
                wd = step.flips.lookup(wk, cout, kh, kw, cin, not weight.requires_grad) \
                    if (step.flips is not None and persistent) else None
                if wd is None:
                    wd = torch.empty(wk.numel(), device=dy.device, dtype=torch.float32)
                    _lib.call("focr_weight_flip_transpose", _p(wk), _p(wd), cout, kh, kw, cin, _stream())
                dx4 = torch.empty_like(x4)
                rc = _lib.load().focr_linear_masked_fwd(_p(dy4), _p(wd), _p(x4), _p(dx4), dy4.numel() // cout, cout, cin,
                                                        ctypes.c_float(ctx.in_relu_scale), _stream())
                if rc == 0:
                    fused_mask = True
                    step.mark_premasked(dx4)
                elif rc != -2:
                    raise RuntimeError("focr_linear_masked_fwd failed: " + _lib.load().focr_last_error().decode())
            if fused_mask:
                pass
            elif _halo_ok(oh, ow, cout, cin, kh, kw, kh - 1 - ph, kw - 1 - pw):
                # data gradient on the halo kernel: flipped weights in fragment order; a single bf16 product under
                # precision mode 3 (csrc/focr_core.hip), split products otherwise
                wf = _frag_weights(step, weight, wk, cout, kh, kw, cin, True)
                # the input was a relu output with THIS layer as its only gradient source (conv2d(fuse_input_relu=True): a
                # parked shortcut gradient, if any, arrives as radd): its relu backward rides in this launch's epilogue
                msk = x4 if (ctx.in_relu_scale == 1.0 and x4.is_contiguous() and _HALO_MASK) else None
                dx4 = _conv_fwd_raw(dy4, None, None, radd, cin, kh, kw, 1, 1, alpha, False, frag=wf,
                                    planes=1 if _lib.get_precision() == 3 else 2, mask=msk)
                if msk is not None:
                    step.mark_premasked(dx4)
            else:
                persistent = weight.is_leaf and wk.data_ptr() == weight.data_ptr()
                wd = step.flips.lookup(wk, cout, kh, kw, cin, not weight.requires_grad) \
                    if (step.flips is not None and persistent) else None
                if wd is None:
                    wd = torch.empty(wk.numel(), device=dy.device, dtype=torch.float32)
                    _lib.call("focr_weight_flip_transpose", _p(wk), _p(wd), cout, kh, kw, cin, _stream())
                dx4 = _conv_fwd_raw(dy4, wd, None, radd, cin, kh, kw, kh - 1 - ph, kw - 1 - pw, alpha, False)
            dx = dx4 if len(ctx.x_shape) == 4 else dx4.reshape(ctx.x_shape)
        if not ctx.needs_input_grad[1] and has_bias and ctx.needs_input_grad[2]:      # bias gradient alone
            db = tb if tb is not None else torch.empty(cout, device=dy.device, dtype=torch.float32)
            _lib.call("focr_colsum", _p(dy4), _p(db), dy4.numel() // cout, cout, cout, _stream())
        if tw is not None:
            dw = None
        if tb is not None:
            db = None
        return dx, dw, db, dres, None, None, None, None, None, None, None, None


_HALO_MASK = os.environ.get("FOCR_HALO_MASK", "1") != "0"      # A/B: relu backward as its own launch


def conv2d(x, weight, bias=None, pad=(0, 0), residual=None, alpha=1.0, relu=False, take_deferred=False,
           defer_residual=False, fuse_input_relu=False):
    """fuse_input_relu: x is the output of a relu `conv2d` AND every gradient of x arrives through THIS call (it is x's
    only consumer, or the other one is a shortcut whose gradient is parked with defer_residual and taken here with
    take_deferred): on a halo-kernel layer the data gradient applies that relu's backward in its epilogue and the
    producer skips its own pass.  Opt-in, as with `linear`: the model code knows the wiring."""
    scale = float(getattr(x, "_focr_relu_scale", 0.0)) if fuse_input_relu else 0.0
    out = _Conv2d.apply(x, weight, bias, residual, pad, alpha, relu, 0.0, take_deferred, defer_residual, False, scale)
    if relu and torch.is_grad_enabled():
        out._focr_relu_scale = 1.0
    return out


def linear(x, weight, bias=None, residual=None, alpha=1.0, relu=False, dropout=0.0, take_deferred=False,
           defer_residual=False, fuse_input_relu=False):
    """x [..., In] @ weight[Out, In]^T (+bias) (+residual) (relu) (dropout with probability `dropout`, relu only).
    take_deferred / defer_residual: see the deferred residual gradients note above.
    fuse_input_relu: x is the output of a relu / relu-dropout `linear` AND THIS CALL IS ITS ONLY CONSUMER (the model
    code knows the wiring, as with the deferred residuals): the data gradient of this layer applies that producer's
    relu backward in its epilogue and the producer's backward skips its own pass.  With a second consumer autograd
    would sum a masked and an unmasked gradient and the producer would scale twice -- hence opt-in, never inferred."""
    scale = float(getattr(x, "_focr_relu_scale", 0.0)) if fuse_input_relu else 0.0
    out = _Conv2d.apply(x, weight, bias, residual, (0, 0), alpha, relu, float(dropout), take_deferred,
                        defer_residual, False, scale)
    if relu and residual is None and torch.is_grad_enabled():
        # the output carries its 1 / P(keep) (the float32 value focr_linear_relu_dropout_fwd reports) for a consumer
        # that opts in with fuse_input_relu
        kq = 65536 - int(float(dropout) * 65536.0 + 0.5) if dropout > 0 else 65536
        out._focr_relu_scale = float(np.float32(65536.0) / np.float32(kq))
    return out


# ----------------------------------------------------------------------------------------
# batch norm (+activation, +residual)
# ----------------------------------------------------------------------------------------
def _row_pitch(t):
    """row pitch (in elements) of a tensor that is a column slice of a contiguous row-major matrix: last dim dense, all
    leading dims collapse to ONE uniform row stride, 16-byte aligned; 0 if it is not (or is plainly contiguous)"""
    if t.dim() < 2 or t.stride(-1) != 1 or t.is_contiguous():
        return 0
    pitch = t.stride(-2)
    if pitch < t.shape[-1] or pitch % 4 or t.data_ptr() % 16:
        return 0
    for d in range(t.dim() - 2):
        if t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
            return 0
    return pitch


_EVAL_INVSTD = {}     # id(running_var) -> ((version, data_ptr, eps, C), invstd, weakref): eval-mode BatchNorm, see forward


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, nbt, residual, training, act, momentum, eps, stats=None):
        c = x.shape[-1]
        rows = x.numel() // c
        _chk(x, gamma, beta, rmean, rvar, residual)
        y = torch.empty_like(x)
        if training:
            # the train kernels rewrite running_mean / running_var through raw pointers (no version bump): a cached
            # eval-mode 1/sqrt(var + eps) of this tensor is stale from here on
            _EVAL_INVSTD.pop(id(rvar), None)
        if training and stats is not None:
            # batch statistics from the producing convolution's per-tile partial sums: no statistics pass over x
            mean = torch.empty(c, device=x.device)
            invstd = torch.empty(c, device=x.device)
            _lib.call("focr_bn_train_fwd_stats", _p(x), _p(stats), stats.shape[0], _p(gamma), _p(beta), _p(rmean),
                      _p(rvar), _p(nbt), _p(residual), _p(y), _p(mean), _p(invstd), rows, c, float(momentum),
                      float(eps), act, _stream())
        elif training:
            mean = torch.empty(c, device=x.device)
            invstd = torch.empty(c, device=x.device)
            ws = torch.empty(_lib.load().focr_bn_ws_floats(rows, c), device=x.device)
            _lib.call("focr_bn_train_fwd", _p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(nbt),
                      _p(residual), _p(y), _p(mean), _p(invstd), _p(ws), rows, c, float(momentum), float(eps),
                      act, _stream())
        else:
            mean = rmean
            # frozen statistics (the recognizer of the training step): 1 / sqrt(var + eps) is computed once per
            # (running_var storage, version, eps) and reused until a train-mode forward on the same tensor drops it
            key = id(rvar)
            hit = _EVAL_INVSTD.get(key)
            if hit is not None and hit[2]() is rvar and hit[0] == (rvar._version, rvar.data_ptr(), float(eps), c):
                invstd = hit[1]
                _lib.call("focr_bn_eval_apply", _p(x), _p(gamma), _p(beta), _p(rmean), _p(invstd), _p(residual), _p(y),
                          rows, c, act, _stream())
            else:
                invstd = torch.empty(c, device=x.device)
                _lib.call("focr_bn_eval_fwd", _p(x), _p(gamma), _p(beta), _p(rmean), _p(rvar), _p(residual), _p(y),
                          _p(invstd), rows, c, float(eps), act, _stream())
                if len(_EVAL_INVSTD) > 256:        # (entries of tensors that are gone)
                    for k_ in [k_ for k_, v_ in _EVAL_INVSTD.items() if v_[2]() is None]:
                        del _EVAL_INVSTD[k_]
                _EVAL_INVSTD[key] = ((rvar._version, rvar.data_ptr(), float(eps), c), invstd, weakref.ref(rvar))
        ctx.cfg = (rows, c, act, bool(training), residual is not None)
        ctx.targets = (_target(gamma), _target(beta))
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dz):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        rows, c, act, training, has_res = ctx.cfg
        # dz may arrive as the feature half of a [.., feature | PE] gradient (kernels._ConcatPE): the kernels take its
        # row pitch instead of a gathered copy
        lddz = _row_pitch(dz) if not has_res else 0
        if lddz == 0:
            dz = dz.contiguous()
        dx = torch.empty_like(x)
        dg = db = None
        if training:
            tg, tb = ctx.targets
            dg = tg if tg is not None else torch.empty(c, device=x.device)
            db = tb if tb is not None else torch.empty(c, device=x.device)
            ws = torch.empty(_lib.load().focr_bn_bwd_ws_floats(rows, c), device=x.device)
            _lib.call("focr_bn_bwd", _p(dz), _p(x), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dx), _p(dg),
                      _p(db), _p(ws), rows, c, act, 1, lddz, _stream())
            dg = None if tg is not None else dg
            db = None if tb is not None else db
        elif ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:       # eval statistics, trainable affine parameters
            tg, tb = ctx.targets
            dg = tg if tg is not None else torch.empty(c, device=x.device)
            db = tb if tb is not None else torch.empty(c, device=x.device)
            ws = torch.empty(_lib.load().focr_bn_bwd_ws_floats(rows, c), device=x.device)
            _lib.call("focr_bn_bwd", _p(dz), _p(x), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dx), _p(dg),
                      _p(db), _p(ws), rows, c, act, 0, lddz, _stream())
            dg = None if tg is not None else dg
            db = None if tb is not None else db
        else:
            _lib.call("focr_bn_bwd", _p(dz), _p(x), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dx), _NULL,
                      _NULL, _NULL, rows, c, act, 0, lddz, _stream())
        return dx, dg, db, None, None, None, (dz if has_res else None), None, None, None, None, None


def batchnorm_act(x, gamma, beta, rmean, rvar, nbt, training, act=ACT_NONE, residual=None,
                  momentum=0.1, eps=1e-5, stats=None):
    return _BatchNormAct.apply(x, gamma, beta, rmean, rvar, nbt, residual, training, act, momentum, eps, stats)


_FOLD_BN = os.environ.get("FOCR_FOLD_BN", "1") != "0"


def conv_bn_foldable(conv, bn, x):
    """conv -> eval-mode BatchNorm with nothing to train (the frozen recognizers of the text- / stroke-focus losses,
    loss/text_focus_loss.py:54-60): the normalisation is a per-channel affine map of the convolution's output and can be
    folded into its weights and bias (as model/crnn/crnn.py does for the frozen CRNN)"""
    if not _FOLD_BN or bn.training or not bn.track_running_stats or not x.is_cuda or conv.weight.dim() != 4:
        return False
    return not any(p is not None and p.requires_grad for p in (conv.weight, conv.bias, bn.weight, bn.bias))


def _conv_bn_folded(conv, bn):
    """(weight * a[co], (bias - mean) * a + beta), a = gamma / sqrt(running_var + eps); cached on the BatchNorm module until
    one of the six tensors changes (address + version counter).  Replaced folds are kept alive: their addresses key the
    frozen entries of FlipTable / FragTable and must never be reused by a newer fold."""
    ts = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts) + (bn.eps,)
    hit = bn.__dict__.get("_focr_fold")
    if hit is None or hit[0] != key:
        with torch.no_grad():
            a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            wf = (conv.weight * a.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
            b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
            bf = ((b0 - bn.running_mean) * a + bn.bias).contiguous()
        if hit is not None:
            bn.__dict__.setdefault("_focr_fold_retired", []).append(hit[1:])
        bn.__dict__["_focr_fold"] = hit = (key, wf, bf)
    return hit[1], hit[2]


def conv_bn(x, conv, bn, act=ACT_NONE, residual=None, take_deferred=False, relu_out=False, defer_residual=False,
            fuse_input_relu=False):
    """act(bn(conv(x))) [+ residual] for a Conv2d / BatchNorm2d module pair (tbsrn.py:246-249, tsrn.py:89-93).  In
    training mode on a halo-kernel layer the convolution's epilogue emits the per-tile sums the BatchNorm needs, so
    the two statistics passes over the conv output disappear.
    relu_out (foldable layers only, see conv_bn_foldable): relu(bn(conv(x)) + residual) -- the tail of a ResNet basic
    block -- as ONE convolution launch with folded weights, residual and relu in its epilogue."""
    if conv_bn_foldable(conv, bn, x) and (relu_out or residual is None) and act in (ACT_NONE, ACT_RELU):
        wf, bf = _conv_bn_folded(conv, bn)
        return conv2d(x, wf, bf, pad=conv.padding, residual=residual, relu=bool(relu_out) or act == ACT_RELU,
                      take_deferred=take_deferred, defer_residual=defer_residual, fuse_input_relu=fuse_input_relu)
    if relu_out:
        raise RuntimeError("conv_bn(relu_out=True) needs a foldable layer (check conv_bn_foldable first)")
    training = bn.training or not bn.track_running_stats
    w = conv.weight
    kh, kw = w.shape[2], w.shape[3]
    stats = None
    if training and x.is_cuda and x.dim() == 4 and _halo_ok(x.shape[1], x.shape[2], x.shape[3], w.shape[0], kh, kw,
                                                            conv.padding[0], conv.padding[1]):
        y, stats = _Conv2d.apply(x, w, conv.bias, None, conv.padding, 1.0, False, 0.0, take_deferred, False, True)
    else:
        y = _Conv2d.apply(x, w, conv.bias, None, conv.padding, 1.0, False, 0.0, take_deferred, False)
    return batchnorm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                         bn.num_batches_tracked if training else None, training, act=act, residual=residual,
                         momentum=bn.momentum, eps=bn.eps, stats=stats)


# ----------------------------------------------------------------------------------------
# the reference's LayerNorm (unbiased std, eps on std), fused residual add
# ----------------------------------------------------------------------------------------
class _LayerNormStd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, a, b, eps, defer=False):
        ctx.step = current_context()
        ctx.defer = bool(defer) and residual is not None
        d = x.shape[-1]
        rows = x.numel() // d
        _chk(x, residual, a, b)
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device)
        rinv = torch.empty(rows, device=x.device)
        _lib.call("focr_layernorm_fwd", _p(x), _p(residual), _p(a), _p(b), _p(y), _p(mean), _p(rinv), rows, d,
                  float(eps), _stream())
        ctx.cfg = (rows, d, float(eps), residual is not None)
        ctx.targets = (_target(a), _target(b))
        ctx.save_for_backward(x, residual, a, mean, rinv)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, residual, a, mean, rinv = ctx.saved_tensors
        rows, d, eps, has_res = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        ta, tb = ctx.targets
        da = ta if ta is not None else torch.empty(d, device=x.device)
        db = tb if tb is not None else torch.empty(d, device=x.device)
        _lib.call("focr_layernorm_bwd", _p(dy), _p(x), _p(residual), _p(a), _p(mean), _p(rinv), _p(dx), _p(da),
                  _p(db), rows, d, eps, int(ta is not None and tb is not None), _stream())
        dres = dx if has_res else None
        if ctx.defer and ctx.needs_input_grad[1]:
            ctx.step.defer_grad(residual, dx)     # added by the data-gradient kernel of the residual's other consumer
            dres = None
        return dx, dres, (None if ta is not None else da), (None if tb is not None else db), None, None


def layernorm_std(x, a, b, residual=None, eps=1e-6, defer=False):
    return _LayerNormStd.apply(x, residual, a, b, eps, defer)


# ----------------------------------------------------------------------------------------
# PReLU / pixel-shuffle+mish / layout + tanh
# ----------------------------------------------------------------------------------------
class _PReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        _chk(x, slope)
        y = torch.empty_like(x)
        _lib.call("focr_prelu_fwd", _p(x), _p(slope), _p(y), x.numel(), _stream())
        ctx.target = _target(slope)
        ctx.save_for_backward(x, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, slope = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        ds = ctx.target if ctx.target is not None else torch.empty(1, device=x.device)
        _lib.call("focr_prelu_bwd", _p(dy), _p(x), _p(slope), _p(dx), _p(ds), x.numel(),
                  int(ctx.target is not None), _stream())
        return dx, (None if ctx.target is not None else ds)


def prelu(x, slope):
    return _PReLU.apply(x, slope)


class _PixelShuffleMish(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre):
        n, h, w, c4 = pre.shape
        _chk(pre)
        z = torch.empty((n, 2 * h, 2 * w, c4 // 4), device=pre.device)
        _lib.call("focr_pixelshuffle_mish_fwd", _p(pre), _p(z), n, h, w, c4 // 4, _stream())
        ctx.save_for_backward(pre)
        return z

    @staticmethod
    def backward(ctx, dz):
        (pre,) = ctx.saved_tensors
        n, h, w, c4 = pre.shape
        dz = dz.contiguous()
        dpre = torch.empty_like(pre)
        _lib.call("focr_pixelshuffle_mish_bwd", _p(dz), _p(pre), _p(dpre), n, h, w, c4 // 4, _stream())
        return dpre


def pixelshuffle_mish(pre):
    return _PixelShuffleMish.apply(pre)


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        n, c, h, w = x.shape
        x = x.contiguous()
        _chk(x)
        y = torch.empty((n, h, w, c), device=x.device)
        _lib.call("focr_nchw_to_nhwc", _p(x), _p(y), n, c, h * w, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c = dy.shape
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), device=dy.device)
        _lib.call("focr_nhwc_to_nchw", _p(dy), _p(dx), n, c, h * w, 0, _stream())
        return dx


def to_nhwc(x):
    return _ToNHWC.apply(x)


class _ToNCHW(torch.autograd.Function):
    """NHWC -> NCHW, optionally through tanh (the SR network's output non-linearity)."""

    @staticmethod
    def forward(ctx, x, do_tanh):
        n, h, w, c = x.shape
        _chk(x)
        y = torch.empty((n, c, h, w), device=x.device)
        _lib.call("focr_nhwc_to_nchw", _p(x), _p(y), n, c, h * w, int(do_tanh), _stream())
        ctx.do_tanh = bool(do_tanh)
        if do_tanh:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = dy.shape
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, c), device=dy.device)
        if ctx.do_tanh:
            (y,) = ctx.saved_tensors
            _lib.call("focr_tanh_bwd_to_nhwc", _p(dy), _p(y), _p(dx), n, c, h * w, _stream())
        else:
            _lib.call("focr_nchw_to_nhwc", _p(dy), _p(dx), n, c, h * w, _stream())
        return dx, None


def to_nchw(x, tanh=False):
    return _ToNCHW.apply(x, tanh)


# ----------------------------------------------------------------------------------------
# FeatureEnhancer pieces
# ----------------------------------------------------------------------------------------
class _ConcatPE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, pe):
        b, t, cf = feat.shape
        cp = pe.shape[-1]
        _chk(feat, pe)
        tok = torch.empty((b, t, cf + cp), device=feat.device)
        _lib.call("focr_concat_pe", _p(feat), _p(pe), _p(tok), b * t, cf, cp, t, _stream())
        ctx.cfg = (b, t, cf, cp)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        b, t, cf, cp = ctx.cfg
        # the feature half as a VIEW: the consumer (the BatchNorm backward of the block's second convolution) reads it
        # through its row pitch; any other consumer gathers it with its own .contiguous()
        return dtok.contiguous()[:, :, :cf], None


def concat_pe(feat, pe):
    return _ConcatPE.apply(feat, pe)


class _Attention(torch.autograd.Function):
    """softmax(q k^T / sqrt(32)) [dropout] v, 4 heads of 32 inside a [B,T,128] row layout."""

    @staticmethod
    def forward(ctx, q, k, v, heads, p_drop, seed):
        b, t, d = q.shape
        _chk(q, k, v)
        o = torch.empty_like(q)
        lse = torch.empty((b, heads, t), device=q.device)
        mask, ready = current_context().next_mask(b, heads, t, p_drop, q.device) if p_drop > 0 else (None, False)
        scale = 1.0 / math.sqrt(d // heads)
        if ready:
            _lib.call("focr_attention_fwd_premasked", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(mask), b, heads, t, d, d,
                      scale, float(p_drop), _stream())
        else:
            _lib.call("focr_attention_fwd", _p(q), _p(k), _p(v), _p(o), _p(lse), _p(mask), b, heads, t, d, d, scale,
                      float(p_drop), seed, _stream())
        ctx.cfg = (b, heads, t, d, scale, float(p_drop))
        ctx.save_for_backward(q, k, v, o, lse, mask)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, mask = ctx.saved_tensors
        b, heads, t, d, scale, p_drop = ctx.cfg
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        work = torch.empty((b, heads, t), device=q.device)
        _lib.call("focr_attention_bwd", _p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(mask), _p(dq), _p(dk),
                  _p(dv), _p(work), b, heads, t, d, d, scale, p_drop, _stream())
        return dq, dk, dv, None, None, None


class _AttentionPacked(torch.autograd.Function):
    """Same attention on a packed projection qkv [B,T,3*D] (q | k | v column blocks): the kernels read the
    three operands as column slices (row pitch 3*D) and write dq | dk | dv into one packed gradient, so the
    surrounding projection is ONE GEMM forward, ONE dgrad GEMM (K = 3*D) and ONE wgrad."""

    @staticmethod
    def forward(ctx, qkv, heads, p_drop, seed):
        b, t, d3 = qkv.shape
        d = d3 // 3
        _chk(qkv)
        o = torch.empty((b, t, d), device=qkv.device)
        lse = torch.empty((b, heads, t), device=qkv.device)
        mask, ready = current_context().next_mask(b, heads, t, p_drop, qkv.device) if p_drop > 0 else (None, False)
        scale = 1.0 / math.sqrt(d // heads)
        if ready:
            _lib.call("focr_attention_fwd_premasked", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _p(o), _p(lse),
                      _p(mask), b, heads, t, d3, d, scale, float(p_drop), _stream())
        else:
            _lib.call("focr_attention_fwd", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _p(o), _p(lse), _p(mask), b,
                      heads, t, d3, d, scale, float(p_drop), seed, _stream())
        ctx.cfg = (b, heads, t, d, scale, float(p_drop))
        ctx.save_for_backward(qkv, o, lse, mask)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse, mask = ctx.saved_tensors
        b, heads, t, d, scale, p_drop = ctx.cfg
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        work = torch.empty((b, heads, t), device=qkv.device)
        _lib.call("focr_attention_bwd", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _p(o), _p(do), _p(lse), _p(mask),
                  _po(dqkv, 0), _po(dqkv, d), _po(dqkv, 2 * d), _p(work), b, heads, t, 3 * d, d, scale, p_drop,
                  _stream())
        return dqkv, None, None, None


def attention_packed(qkv, heads=4, p_drop=0.0):
    seed = _new_seed() if p_drop > 0 else 0
    return _AttentionPacked.apply(qkv, heads, p_drop, seed)


def attention(q, k, v, heads=4, p_drop=0.0):
    seed = _new_seed() if p_drop > 0 else 0
    return _Attention.apply(q, k, v, heads, p_drop, seed)


# ----------------------------------------------------------------------------------------
# The whole FeatureEnhancer as ONE autograd node (reference tbsrn.py:76-92).  Forward: concat-PE, packed QKV projection,
# attention, then the two fused row chains of csrc/fe_chain.hip; backward: the two data-gradient chains, the attention
# backward, the QKV data gradient restricted to the feature columns, and every parameter gradient of the block in one
# library call on the weight-gradient side stream.  5 + 5 library calls instead of ~10 + ~25 autograd nodes, and the
# intermediates between the row-local layers never touch HBM.
# ----------------------------------------------------------------------------------------
def fe_chain_supported(feat, heads=4, d_model=128):
    """The fused chains (csrc/fe_chain.hip) are written for 4 heads of 32 over d_model = 128 (fe_bwd_b's accumulator
    tile j IS head j of D = rowsum(dO * O)); any other geometry takes the per-layer path."""
    return bool(int(heads) == 4 and int(d_model) == 128
                and feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 3 and feat.shape[-1] == 64
                and _lib.load().focr_fe_chain_supported(feat.shape[0] * feat.shape[1], 128))


# FOCR_C9_WGRAD_BX3=0: weight gradient of the 9x9 output layer on the round-1 fp32-MFMA kernel (A/B switch)
_C9_WGRAD_BX3 = os.environ.get("FOCR_C9_WGRAD_BX3", "1") != "0"
_LOG2E = 1.4426950408889634
# FOCR_DGRAD_FIRST=0: enqueue a convolution's weight gradient (side stream) before its data gradient (main stream)
_DGRAD_FIRST = os.environ.get("FOCR_DGRAD_FIRST", "1") != "0"
# FOCR_DEFER_SIDE=1: park the convolution / QKV weight gradients of a residual block and issue them beside the NEXT
# block's attention backward.  Measured (same box, interleaved): 15.32 ms vs 15.13 ms without -- the attention kernels lose
# more to the extra company than the HBM-bound kernels gain; off by default, kept as an A/B switch.
_DEFER_SIDE = os.environ.get("FOCR_DEFER_SIDE", "0") == "1"
# FOCR_PARK_TAIL=k: weight gradients of the k residual blocks that come last in backward order wait for the STN head's
# backward (StepContext.park_tail).  Measured (tools/gpu/r04_call30.sh, interleaved, two rounds): k = 0 / 1 / 2 / 3 ->
# 13.50 / 13.56 / 13.79 / 13.99 ms -- the tail (0.5 ms of small launches) is shorter than one block's weight gradients
# (0.25 ms each, un-overlapped), which the default schedule already hides beside that block's own kernels; off by default.
_PARK_TAIL = int(os.environ.get("FOCR_PARK_TAIL", "0"))
# FOCR_BN2_FUSE=0: the residual block's second BatchNorm as its own apply pass (A/B switch)
_BN2_FUSE = os.environ.get("FOCR_BN2_FUSE", "1") != "0"
# FOCR_MASK_EARLY=1: next step's attention keep bits drawn under the recognizer's LSTM scan (second buffer per attention
# call) instead of at the step start.  Measured (tools/gpu/r04_call25.sh, interleaved): 13.56 / 13.55 ms with, 13.45 / 13.50
# ms without -- the 8-block groups of the persistent scan lose more to the company of the mask blocks than the first
# forward kernels of the step gain; off by default, kept as an A/B switch.
_MASK_EARLY = os.environ.get("FOCR_MASK_EARLY", "0") == "1"
# FOCR_FE_WGRAD_EARLY=0: all weight gradients of a FeatureEnhancer after its attention backward (A/B measurements)
_FE_WGRAD_EARLY = os.environ.get("FOCR_FE_WGRAD_EARLY", "1") != "0"
FE_PARAM_NAMES = ("wqkv", "bqkv", "wo", "bo", "a1", "b1", "w1", "bb1", "w2", "bb2", "a3", "b3", "wl", "bl")


def _fe_forward(step, feat, xres, pe, heads, p_attn, p_ffn, eps, params, bn=None):
    """forward of the block on tokens feat [B, T, 64] -> (out [B, T, 64], saved tensors, cfg).
    bn = (gamma, beta, mean, invstd): feat is the INPUT of the BatchNorm in front of the block; the projection kernel
    normalises it on load (focr_fe_qkv_fwd_bn)"""
    wqkv, bqkv, wo, bo, a1, b1, w1, bb1, w2, bb2, a3, b3, wl, bl = params
    step._fe_fwd_n += 1
    b, t, cf = feat.shape
    rows, d = b * t, 128
    dev = feat.device
    tok = torch.empty((b, t, d), device=dev)
    if pe.shape[0] != t or pe.shape[-1] != 64:
        raise RuntimeError("positional-encoding table must be [tokens per image, 64]")
    o = torch.empty((b, t, d), device=dev)
    lse = torch.empty((b, heads, t), device=dev)
    mask, ready = step.next_mask(b, heads, t, p_attn, dev) if p_attn > 0 else (None, False)
    scale = 1.0 / math.sqrt(d // heads)
    qkv = torch.empty((b, t, 3 * d), device=dev)
    _lib.call("focr_fe_qkv_fwd_bn", _p(feat), _p(pe), _p(wqkv), _p(bqkv), _p(tok), _p(qkv), rows, t, _NULL, 1.0,
              *[_p(v_) for v_ in (bn or (None,) * 4)], _stream())
    if ready:
        _lib.call("focr_attention_fwd_premasked", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _p(o), _p(lse),
                  _p(mask), b, heads, t, 3 * d, d, scale, float(p_attn), _stream())
    else:
        _lib.call("focr_attention_fwd", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _p(o), _p(lse), _p(mask), b,
                  heads, t, 3 * d, d, scale, float(p_attn), _new_seed() if p_attn > 0 else 0, _stream())
    xhat1, xhat2, h = torch.empty_like(tok), torch.empty_like(tok), torch.empty_like(tok)
    rinv1, rinv2 = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    out = torch.empty((b, t, cf), device=dev)
    ks = ctypes.c_float(1.0)
    _lib.call("focr_fe_post_fwd", _p(o), _p(tok), _p(xres), _p(wo), _p(bo), _p(a1), _p(b1), _p(w1), _p(bb1),
              _p(w2), _p(bb2), _p(a3), _p(b3), _p(wl), _p(bl), _p(xhat1), _p(rinv1), _p(h), _p(xhat2), _p(rinv2),
              _p(out), rows, float(eps), float(p_ffn), _new_seed() if p_ffn > 0 else 0, ctypes.byref(ks), _stream())
    cfg = (b, t, heads, scale, float(p_attn), float(eps), float(ks.value))
    return out, (tok, qkv, o, lse, mask, xhat1, rinv1, h, xhat2, rinv2), cfg


def _fe_backward(step, saved, cfg, params, targets, d_out, need_dfeat, need_params, clone_for_side=None):
    """backward of the block: -> (d_feat or None, parameter gradients (None where a flat-buffer target took them)).
    d_out [B, T, 64] is ALSO the gradient of the block's residual input; the caller routes it.  clone_for_side: callable
    invoked when the weight gradients go to the side stream (the caller may have to protect d_out from in-place reuse).

    Weight gradients in two side-stream calls: the four linears whose operands the chain kernels produce are issued
    BEFORE the attention backward -- those streaming kernels (0.5 GB of operands) then share the chip with the VALU-bound
    attention kernels instead of with the HBM-bound BatchNorm / convolution data-gradient kernels that follow; only the
    packed q | k | v projection has to wait for dqkv."""
    tok, qkv, o, lse, mask, xhat1, rinv1, h, xhat2, rinv2 = saved
    wqkv, bqkv, wo, bo, a1, b1, w1, bb1, w2, bb2, a3, b3, wl, bl = params
    b, t, heads, scale, p_attn, eps, keep_scale = cfg
    rows, d = b * t, 128
    dev = d_out.device
    d_s2, d_hpre, d_s1 = (torch.empty_like(tok) for _ in range(3))
    work = torch.empty((b, heads, t), device=dev)
    d_ctx = torch.empty_like(tok)
    _lib.call("focr_fe_post_bwd", _p(d_out), _p(wl), _p(xhat2), _p(rinv2), _p(a3), _p(w2), _p(h), keep_scale,
              _p(w1), _p(xhat1), _p(rinv1), _p(a1), _p(wo), _p(d_s2), _p(d_hpre), _p(d_s1), _p(d_ctx), rows, eps,
              _p(o), _p(work), t, _NULL, 1.0, _stream())
    grads = [None] * len(FE_PARAM_NAMES)
    side, g, nws = None, None, 0
    if need_params:
        tg = list(targets)
        flat = all(x is not None for x in tg)
        if not flat:
            tg = [torch.empty_like(p_, memory_format=torch.contiguous_format) for p_ in params]
            grads = list(tg)
        side = step.side_stream() if flat else None
        nws = _lib.load().focr_fe_wgrads_ws_floats(rows)
        g = dict(zip(FE_PARAM_NAMES, tg))
        if side is not None and clone_for_side is not None:
            clone_for_side()

    def wgrads(parts, used):
        if side is not None:
            ev = torch.cuda.Event()
            ev.record()
            side.wait_event(ev)
            for x in used:
                x.record_stream(side)
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            ws = torch.empty(nws, device=dev)
            _lib.call("focr_fe_wgrads", _p(d_out), _p(xhat2), _p(d_s2), _p(h), _p(d_hpre), _p(xhat1), _p(d_s1),
                      _p(o), _p(dqkv), _p(tok), _p(wl), _p(w1), _p(a1), _p(b1), _p(a3), _p(b3), _p(g["wl"]),
                      _p(g["bl"]), _p(g["a3"]), _p(g["b3"]), _p(g["w2"]), _p(g["bb2"]), _p(g["w1"]), _p(g["bb1"]),
                      _p(g["a1"]), _p(g["b1"]), _p(g["wo"]), _p(g["bo"]), _p(g["wqkv"]), _p(g["bqkv"]), _p(ws),
                      nws, rows, parts, _stream())
    dqkv = None
    step.flush_side()          # the previous block's parked weight gradients: beside THIS block's attention backward
    step.fe_backward_begins()
    if need_params and _FE_WGRAD_EARLY:
        step.park_tail(lambda: wgrads(1, (d_out, xhat2, d_s2, h, d_hpre, xhat1, d_s1, o)))
    dqkv = torch.empty((b, t, 3 * d), device=dev)
    # `work` already holds D = rowsum(d_ctx * o) per (b, head, token), written by the chain kernel above
    _lib.call("focr_attention_bwd", _po(qkv, 0), _po(qkv, d), _po(qkv, 2 * d), _NULL, _p(d_ctx), _p(lse), _p(mask),
              _po(dqkv, 0), _po(dqkv, d), _po(dqkv, 2 * d), _p(work), b, heads, t, 3 * d, d, scale, p_attn,
              _stream())
    d_feat = None
    if need_dfeat:
        d_feat = torch.empty((b, t, 64), device=dev)
        _lib.call("focr_fe_qkv_dgrad", _p(dqkv), _p(wqkv), _p(d_s1), _p(d_feat), rows, _stream())
    if need_params:
        if _FE_WGRAD_EARLY:
            step.park_tail(lambda: wgrads(2, (dqkv, tok)), step.defer_side)
        else:
            step.park_tail(lambda: wgrads(3, (d_out, xhat2, d_s2, h, d_hpre, xhat1, d_s1, o, dqkv, tok)), step.defer_side)
    return d_feat, grads


class _FeatureEnhancerFused(torch.autograd.Function):
    N_PARAMS = len(FE_PARAM_NAMES)

    @staticmethod
    def forward(ctx, feat, xres, pe, heads, p_attn, p_ffn, eps, defer_residual, *params):
        ctx.step = step = current_context()
        _chk(feat, xres, pe, *params)
        out, saved, ctx.cfg = _fe_forward(step, feat, xres, pe, heads, p_attn, p_ffn, eps, params)
        ctx.defer_residual = bool(defer_residual) and xres is not None
        ctx.res_key = _dkey(xres) if ctx.defer_residual else None
        ctx.targets = tuple(_target(p_) for p_ in params)
        ctx.save_for_backward(*saved, *params)
        return out

    @staticmethod
    def backward(ctx, dy):
        saved, params = ctx.saved_tensors[:10], ctx.saved_tensors[10:]
        step = ctx.step
        d_out = dy.contiguous()
        # residual gradient of the block input: parked for the data-gradient kernel of its other consumer (the block's
        # first convolution), or handed to autograd
        box = {"dres": None}
        if ctx.needs_input_grad[1]:
            if ctx.defer_residual:
                if ctx.res_key in step.deferred:
                    raise RuntimeError("two deferred gradients for the same tensor")
                step.deferred[ctx.res_key] = d_out
            else:
                box["dres"] = d_out

        def protect():      # autograd may accumulate into dres in place while the side stream still reads d_out
            if box["dres"] is not None:
                box["dres"] = box["dres"].clone()
        d_feat, grads = _fe_backward(step, saved, ctx.cfg, params, ctx.targets, d_out, ctx.needs_input_grad[0],
                                     any(ctx.needs_input_grad[8:]), protect)
        return (d_feat, box["dres"], None, None, None, None, None, None) + tuple(grads)


def feature_enhancer_fused(feat, xres, pe, params, heads=4, p_attn=0.0, p_ffn=0.0, eps=1e-6, defer_residual=False):
    """params: (wqkv [384,128], bqkv, wo, bo, a1, b1, w1, bb1, w2, bb2, a3, b3, wl [64,128], bl) -- tbsrn.py:76-92"""
    if int(heads) != 4 or tuple(params[0].shape) != (384, 128):
        raise RuntimeError("feature_enhancer_fused: the fused chains are built for 4 heads of 32 (d_model 128); "
                           "got heads=%r, wqkv %r -- use the per-layer path (fe_chain_supported)" % (heads, tuple(params[0].shape)))
    return _FeatureEnhancerFused.apply(feat, xres, pe, int(heads), float(p_attn), float(p_ffn), float(eps),
                                       bool(defer_residual), *params)


# ----------------------------------------------------------------------------------------
# A whole TBSRN residual block (reference tbsrn.py:246-257) as ONE autograd node in training mode:
#   conv3x3 -> BatchNorm(batch statistics) -> mish -> conv3x3 -> BatchNorm -> FeatureEnhancer (+ block input).
# Same library calls as the per-layer nodes (halo convolution with the BatchNorm partial sums in its epilogue, statistics
# fold + apply, the fused FeatureEnhancer chains), but ~13 forward + ~17 backward calls are issued from two Python
# functions instead of five autograd nodes each way: the host side of the step is what bounds the small-batch
# configurations.  The block input's residual gradient rides in the epilogue of the first convolution's data gradient.
# ----------------------------------------------------------------------------------------
def srb_fused_supported(x, conv1, conv2, bn1, bn2):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[-1] == 64 and x.is_contiguous()):
        return False
    n, h, w, c = x.shape
    for cv in (conv1, conv2):
        if tuple(cv.weight.shape) != (64, 64, 3, 3) or tuple(cv.padding) != (1, 1) or cv.bias is None:
            return False
        if _ohwi(cv.weight).data_ptr() != cv.weight.data_ptr():
            return False          # weights re-laid-out by the caller (not channels_last): the per-layer path copies them
    if not all((bn.training or not bn.track_running_stats) and bn.track_running_stats for bn in (bn1, bn2)):
        return False
    if any(bn.momentum is None for bn in (bn1, bn2)):
        return False              # cumulative-average running statistics: per-layer path
    return bool(_halo_ok(h, w, 64, 64, 3, 3, 1, 1) and _lib.load().focr_fe_chain_supported(n * h * w, 128))


class _SRBFused(torch.autograd.Function):
    N_CONV = 8           # w1 c1 g1 be1 w2 c2 g2 be2 (conv weight / bias, BatchNorm weight / bias, twice)

    @staticmethod
    def forward(ctx, x, pe, heads, p_attn, p_ffn, eps_ln, bn_cfg, bn_bufs, *params):
        """bn_cfg: ((momentum1, eps1), (momentum2, eps2)); bn_bufs: running_mean / running_var / num_batches_tracked of
        the two BatchNorms (updated in place, as nn.BatchNorm2d does); params: the 8 conv / BatchNorm tensors, then the
        14 FeatureEnhancer tensors (FE_PARAM_NAMES)"""
        ctx.step = step = current_context()
        cparams, fparams = params[:_SRBFused.N_CONV], params[_SRBFused.N_CONV:]
        _chk(x, pe, *[_ohwi(p_) if p_.dim() == 4 else p_ for p_ in params])
        if any(_ohwi(p_).data_ptr() != p_.data_ptr() for p_ in (cparams[0], cparams[4])):
            raise RuntimeError("the block's convolution weights must be channels_last tensors")
        n, h, w, c = x.shape
        rows = n * h * w
        dev = x.device
        lib = _lib.load()
        tiles = lib.focr_conv3x3_frag_tiles(n, h, w)
        ys, zs, means, invs = [], [], [], []
        inp = x
        for i in range(2):
            wgt, bias, gamma, beta = cparams[4 * i:4 * i + 4]
            rmean, rvar, nbt = bn_bufs[3 * i:3 * i + 3]
            _EVAL_INVSTD.pop(id(rvar), None)          # running statistics are rewritten below (see _BatchNormAct)
            mom, eps_bn = bn_cfg[i]
            frag = _frag_weights(step, wgt, _ohwi(wgt), 64, 3, 3, 64, False)
            y = torch.empty((n, h, w, 64), device=dev)
            stats = torch.empty((tiles, 64, 2), device=dev)
            _lib.call("focr_conv3x3_frag_fwd", _p(inp), ctypes.c_void_p(frag.data_ptr()), _p(bias), _NULL, _p(y),
                      _p(stats), n, h, w, 64, 64, 1.0, 0, 2, 0, 0, 0, _stream())
            # the second BatchNorm's output has ONE consumer, the packed projection's load: statistics only here, the
            # normalisation rides on that load (FOCR_BN2_FUSE=0: separate apply pass)
            fuse = i == 1 and _BN2_FUSE
            z = None if fuse else torch.empty_like(y)
            mean, invstd = torch.empty(64, device=dev), torch.empty(64, device=dev)
            _lib.call("focr_bn_train_fwd_stats", _p(y), _p(stats), tiles, _p(gamma), _p(beta), _p(rmean), _p(rvar),
                      _p(nbt), _NULL, _p(z), _p(mean), _p(invstd), rows, 64, float(mom), float(eps_bn),
                      ACT_MISH if i == 0 else ACT_NONE, _stream())
            ys.append(y), zs.append(z), means.append(mean), invs.append(invstd)
            inp = z
        if zs[1] is None:
            out, fsaved, ctx.cfg = _fe_forward(step, ys[1].view(n, h * w, 64), x.view(n, h * w, 64), pe, heads, p_attn,
                                               p_ffn, eps_ln, fparams, bn=(cparams[6], cparams[7], means[1], invs[1]))
        else:
            out, fsaved, ctx.cfg = _fe_forward(step, zs[1].view(n, h * w, 64), x.view(n, h * w, 64), pe, heads, p_attn,
                                               p_ffn, eps_ln, fparams)
        ctx.geom = (n, h, w)
        ctx.targets = tuple(_target(p_) for p_ in params)
        ctx.save_for_backward(x, ys[0], zs[0], ys[1], means[0], invs[0], means[1], invs[1], *fsaved, *params)
        return out.view(n, h, w, 64)

    @staticmethod
    def backward(ctx, dy):
        t = ctx.saved_tensors
        x, y1, z1, y2, mean1, inv1, mean2, inv2 = t[:8]
        fsaved, params = t[8:18], t[18:]
        cparams, fparams = params[:_SRBFused.N_CONV], params[_SRBFused.N_CONV:]
        step = ctx.step
        n, h, w = ctx.geom
        rows = n * h * w
        dev = dy.device
        lib = _lib.load()
        d_out = dy.contiguous().view(n, h * w, 64)
        need_p = any(ctx.needs_input_grad[8:])
        d_feat, fgrads = _fe_backward(step, fsaved, ctx.cfg, fparams, ctx.targets[_SRBFused.N_CONV:], d_out, True,
                                      need_p)
        grads = [None] * _SRBFused.N_CONV
        planes = 1 if _lib.get_precision() == 3 else 2
        dz = d_feat.view(n, h, w, 64)
        dx = None
        for i in (1, 0):
            wgt, bias, gamma, beta = cparams[4 * i:4 * i + 4]
            tw, tb, tg, tbe = ctx.targets[4 * i:4 * i + 4]
            yy, mean, invstd = (y2, mean2, inv2) if i == 1 else (y1, mean1, inv1)
            cin = z1 if i == 1 else x
            # BatchNorm backward (batch statistics): dgamma / dbeta straight into their targets, then d(conv output)
            dg = tg if tg is not None else torch.empty(64, device=dev)
            db = tbe if tbe is not None else torch.empty(64, device=dev)
            ws = torch.empty(lib.focr_bn_bwd_ws_floats(rows, 64), device=dev)
            dyc = torch.empty_like(yy)
            _lib.call("focr_bn_bwd", _p(dz), _p(yy), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(dyc), _p(dg), _p(db),
                      _p(ws), rows, 64, ACT_MISH if i == 0 else ACT_NONE, 1, 0, _stream())
            if tg is None:
                grads[4 * i + 2] = dg
            if tbe is None:
                grads[4 * i + 3] = db
            # the data gradient is on the step's critical path, the weight gradient is not: the main-stream kernel is
            # enqueued FIRST (the side stream only needs dyc, marked by an event recorded before it), so it gets the CUs first
            ev_dy = None
            if _DGRAD_FIRST and need_p and step.side_enabled and not _DEFER_SIDE and not step._parking:
                ev_dy = torch.cuda.Event()
                ev_dy.record()
            if _DGRAD_FIRST:
                # data gradient on the halo kernel (flipped fragment weights); the block input's residual gradient (= d_out)
                # is added in the epilogue of the FIRST convolution's data gradient
                wf = _frag_weights(step, wgt, _ohwi(wgt), 64, 3, 3, 64, True)
                dzn = torch.empty((n, h, w, 64), device=dev)
                _lib.call("focr_conv3x3_frag_fwd", _p(dyc), ctypes.c_void_p(wf.data_ptr()), _NULL,
                          _p(d_out) if i == 0 else _NULL, _p(dzn), _NULL, n, h, w, 64, 64, 1.0, 0, planes, 0, 0, 0, _stream())
            # convolution weight / bias gradient: side stream when the targets are flat-buffer slices
            if need_p:
                flat = tw is not None and tb is not None
                dw = tw if flat else torch.empty((64, 3, 3, 64), device=dev).permute(0, 3, 1, 2)
                dbias = tb if flat else torch.empty(64, device=dev)

                def conv_wgrad(cin=cin, dyc=dyc, dw=dw, dbias=dbias, flat=flat, ev=ev_dy):
                    side = step.side_stream() if flat else None
                    if side is not None:
                        if ev is None:             # (also when parked: everything up to the flush point)
                            ev = torch.cuda.Event()
                            ev.record()
                        side.wait_event(ev)
                        cin.record_stream(side)
                        dyc.record_stream(side)
                    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                        nws = lib.focr_conv2d_wgrad_ws_floats(n, h, w, 64, 64, 3, 3, 1, 1)
                        wsw = torch.empty(nws, device=dev) if nws > 0 else None
                        _lib.call("focr_conv2d_wgrad", _p(cin), _p(dyc), _p(dw), _p(dbias), n, h, w, 64, 64, 3, 3, 1, 1,
                                  0, 0, int(flat), _p(wsw), nws, _stream())
                if flat:
                    step.park_tail(conv_wgrad, step.defer_side)   # (defer_side: beside the next block's attention backward)
                else:
                    conv_wgrad()
                    grads[4 * i], grads[4 * i + 1] = dw, dbias
            if not _DGRAD_FIRST:
                # data gradient on the halo kernel (flipped fragment weights); the block input's residual gradient (= d_out)
                # is added in the epilogue of the FIRST convolution's data gradient
                wf = _frag_weights(step, wgt, _ohwi(wgt), 64, 3, 3, 64, True)
                dzn = torch.empty((n, h, w, 64), device=dev)
                _lib.call("focr_conv3x3_frag_fwd", _p(dyc), ctypes.c_void_p(wf.data_ptr()), _NULL,
                          _p(d_out) if i == 0 else _NULL, _p(dzn), _NULL, n, h, w, 64, 64, 1.0, 0, planes, 0, 0, 0, _stream())
            dz = dzn
        dx = dz if ctx.needs_input_grad[0] else None
        return (dx, None, None, None, None, None, None, None) + tuple(grads) + tuple(fgrads)


def srb_fused(x, pe, conv1, bn1, conv2, bn2, fe_params, heads=4, p_attn=0.0, p_ffn=0.0, eps_ln=1e-6):
    bufs = (bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, bn2.running_mean, bn2.running_var,
            bn2.num_batches_tracked)
    cfgs = ((bn1.momentum, bn1.eps), (bn2.momentum, bn2.eps))
    return _SRBFused.apply(x, pe, int(heads), float(p_attn), float(p_ffn), float(eps_ln), cfgs, bufs, conv1.weight,
                           conv1.bias, bn1.weight, bn1.bias, conv2.weight, conv2.bias, bn2.weight, bn2.bias, *fe_params)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        _chk(x)
        y = torch.empty_like(x)
        _lib.call("focr_dropout", _p(x), _p(y), x.numel(), float(p), seed, _stream())
        ctx.cfg = (float(p), seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.call("focr_dropout", _p(dy), _p(dx), dy.numel(), p, seed, _stream())
        return dx, None, None


def dropout(x, p, training):
    if not training or p <= 0:
        return x
    return _Dropout.apply(x, p, _new_seed())


# ----------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------
class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        out = torch.empty(1, device=a.device)
        _lib.call("focr_mse_fwd", _p(a), _p(b), _p(out), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous().reshape(1)
        da = torch.empty_like(a)
        _lib.call("focr_mse_bwd", _p(a), _p(b), _p(g), _p(da), a.numel(), _stream())
        return da, None


def mse_loss(a, b):
    return _MSE.apply(a, b)


class _CTC(torch.autograd.Function):
    """log_softmax + CTC (blank 0, reduction 'mean', zero_infinity) on logits [T,B,C]."""

    @staticmethod
    def forward(ctx, logits, targets, lengths, offsets):
        t, b, c = logits.shape
        logits = logits.contiguous()
        _chk(logits)
        loss = torch.empty(1, device=logits.device)
        nll = torch.empty(b, device=logits.device)
        grad = torch.empty_like(logits)
        _lib.call("focr_ctc_fwd", _p(logits), _p(targets), _p(lengths), _p(offsets), _p(loss), _p(nll),
                  _p(grad), t, b, c, _stream())
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        g = g.contiguous().reshape(1)
        out = torch.empty_like(grad)
        _lib.call("focr_scale_dev", _p(grad), _p(g), _p(out), grad.numel(), _stream())
        return out, None, None, None


def ctc_loss(logits, targets, lengths, offsets=None):
    """targets: int32 [sum L] (device), lengths: int32 [B] (device), offsets: int32 [B] exclusive prefix sums of the lengths
    (CTCFocusLoss.encode computes them on the host; derived here when absent)."""
    if offsets is None:
        offsets = (torch.cumsum(lengths, 0) - lengths).to(torch.int32)
    return _CTC.apply(logits, targets.to(torch.int32).contiguous(), lengths.to(torch.int32).contiguous(),
                      offsets.contiguous())


# ----------------------------------------------------------------------------------------
# pooling / resampling
# ----------------------------------------------------------------------------------------
class _MaxPool(torch.autograd.Function):
    """relu_input: x is the output of a conv + relu layer (CRNN conv0 / 1 / 3 / 5).  The backward then applies that relu's
    backward itself (a window whose maximum is 0 passes no gradient) and registers its result in StepContext.premasked,
    where the convolution's backward finds it and skips its own relu pass over the un-pooled gradient."""

    @staticmethod
    def forward(ctx, x, kernel, stride, pad, relu_input):
        n, h, w, c = x.shape
        _chk(x)
        kh, kw = kernel
        sh, sw = stride
        ph, pw = pad
        oh, ow = (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
        y = torch.empty((n, oh, ow, c), device=x.device)
        idx = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        _lib.call("focr_maxpool_fwd", _p(x), _p(y), ctypes.c_void_p(idx.data_ptr()), n, h, w, c, kh, kw, sh, sw,
                  ph, pw, _stream())
        ctx.cfg = (n, h, w, c, kh, kw, sh, sw, ph, pw)
        ctx.relu_input = bool(relu_input) and c % 4 == 0
        ctx.step = current_context()
        if ctx.relu_input:
            ctx.save_for_backward(idx, y)
        else:
            ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx = ctx.saved_tensors[0]
        n, h, w, c, kh, kw, sh, sw, ph, pw = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, c), device=dy.device)
        if ctx.relu_input:
            _lib.call("focr_maxpool_relu_bwd", _p(dy), ctypes.c_void_p(idx.data_ptr()), _p(ctx.saved_tensors[1]), _p(dx), n,
                      h, w, c, kh, kw, sh, sw, ph, pw, _stream())
            ctx.step.mark_premasked(dx)
        else:
            _lib.call("focr_maxpool_bwd", _p(dy), ctypes.c_void_p(idx.data_ptr()), _p(dx), n, h, w, c, kh, kw, sh,
                      sw, ph, pw, _stream())
        return dx, None, None, None, None


def maxpool(x, kernel, stride=None, pad=(0, 0), relu_input=False):
    return _MaxPool.apply(x, tuple(kernel), tuple(stride or kernel), tuple(pad), relu_input)


class _Conv0ReluPool(torch.autograd.Function):
    """Conv2d(1, 64, 3, 1, 1) -> ReLU -> MaxPool2d(2, 2) of a FROZEN recognizer in one launch each way
    (csrc/crnn_conv0_pool.hip; model/crnn/crnn.py:51-52 of the reference): the 134 MB full-resolution activation and its
    gradient never exist.  x: [N, H, W, 1] NHWC; weight [64, 1, 3, 3] and bias get no gradient here -- `conv0_relu_pool`
    falls back to the per-layer path when they need one."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, h, w, _ = x.shape
        _chk(x, weight)
        y = torch.empty((n, h // 2, w // 2, 64), device=x.device)
        idx = torch.empty((n, h // 2, w // 2, 64), device=x.device, dtype=torch.uint8)
        _lib.call("focr_crnn_conv0_pool_fwd", _p(x), _p(weight), _p(bias), _p(y),
                  ctypes.c_void_p(idx.data_ptr()), n, h, w, _stream())
        ctx.cfg = (n, h, w)
        ctx.save_for_backward(idx, y, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, y, weight = ctx.saved_tensors
        n, h, w = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((n, h, w, 1), device=dy.device)
        _lib.call("focr_crnn_conv0_pool_bwd", _p(dy), ctypes.c_void_p(idx.data_ptr()), _p(y), _p(weight), _p(dx), n, h, w,
                  _stream())
        return dx, None, None


# FOCR_CONV0_POOL=0: first recognizer layer on the per-layer path (convolution, pooling, their two backward launches)
_CONV0_POOL = os.environ.get("FOCR_CONV0_POOL", "1") != "0"


def conv0_relu_pool_supported(x, weight, bias, pool_kernel, pool_stride, pool_pad):
    """the fused first layer applies to: NHWC input with one channel, a 3x3 / pad 1 / 64-channel convolution whose
    parameters need no gradient, 2x2 / stride 2 pooling, H % 4 == 0, even W <= 128"""
    if not (_CONV0_POOL and x.is_cuda and x.dim() == 4 and x.shape[-1] == 1 and tuple(weight.shape) == (64, 1, 3, 3)):
        return False
    if torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False
    if tuple(pool_kernel) != (2, 2) or tuple(pool_stride) != (2, 2) or tuple(pool_pad) != (0, 0):
        return False
    if not (weight.is_contiguous() and (bias is None or bias.is_contiguous())):
        return False
    return bool(_lib.load().focr_crnn_conv0_pool_supported(x.shape[1], x.shape[2], 1, 64, 3, 3, 1))


def conv0_relu_pool(x, weight, bias):
    return _Conv0ReluPool.apply(x.contiguous(), weight, bias)


class _TPSWarp(torch.autograd.Function):
    """TPS grid + bilinear sampling of an NHWC image; gradients for the control points and -- when the image itself
    requires one (d loss / d LR image, F.grid_sample's input gradient) -- for the image."""

    @staticmethod
    def forward(ctx, img, ctrl, inv_kernel, coord_repr):
        b, h, w, c = img.shape
        nc = ctrl.shape[1]
        ctrl = ctrl.contiguous()
        _chk(img, ctrl, inv_kernel, coord_repr)
        out = torch.empty_like(img)
        src = torch.empty((b, h * w, 2), device=img.device)
        _lib.call("focr_tps_fwd", _p(img), _p(ctrl), _p(inv_kernel), _p(coord_repr), _p(out), _p(src), b, h, w,
                  c, nc, _stream())
        ctx.cfg = (b, h, w, c, nc)
        ctx.step = current_context()
        if ctx.needs_input_grad[1]:
            ctx.step.tail_ready = True       # a trainable STN head follows in the backward: release point of park_tail
        ctx.save_for_backward(img, src, inv_kernel, coord_repr)
        return out

    @staticmethod
    def backward(ctx, dout):
        img, src, inv_kernel, coord_repr = ctx.saved_tensors
        b, h, w, c, nc = ctx.cfg
        dout = dout.contiguous()
        ctx.step.flush_tail()        # parked weight gradients: beside the STN head's backward (StepContext.park_tail)
        dctrl = torch.empty((b, nc, 2), device=dout.device)
        _lib.call("focr_tps_bwd", _p(dout), _p(img), _p(src), _p(inv_kernel), _p(coord_repr), _p(dctrl), b, h,
                  w, c, nc, _stream())
        dimg = None
        if ctx.needs_input_grad[0]:
            dimg = torch.empty_like(img)
            _lib.call("focr_tps_bwd_img", _p(dout), _p(src), _p(dimg), b, h, w, c, _stream())
        return dimg, dctrl, None, None


def tps_warp(img, ctrl, inv_kernel, coord_repr):
    return _TPSWarp.apply(img, ctrl, inv_kernel, coord_repr)


class _BicubicGray(torch.autograd.Function):
    """parse_crnn_data: NCHW [B,C>=3,H,IW] -> [B,1,H,OW] (bicubic along W, luma)."""

    @staticmethod
    def forward(ctx, x, ow):
        b, c, h, iw = x.shape
        x = x.contiguous()
        _chk(x)
        y = torch.empty((b, 1, h, ow), device=x.device)
        _lib.call("focr_bicubic_gray_fwd", _p(x), _p(y), b, c, h, iw, ow, _stream())
        ctx.cfg = (b, c, h, iw, ow)
        return y

    @staticmethod
    def backward(ctx, dy):
        b, c, h, iw, ow = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((b, c, h, iw), device=dy.device)
        _lib.call("focr_bicubic_gray_bwd", _p(dy), _p(dx), b, c, h, iw, ow, _stream())
        return dx, None


def bicubic_gray(x, ow=100):
    return _BicubicGray.apply(x, ow)


# ----------------------------------------------------------------------------------------
# bidirectional LSTM (recurrent part; the input projection is a `linear`)
# ----------------------------------------------------------------------------------------
def _lstm_check(ws, batch):
    """FOCR_LSTM_CHECK=1 (debugging / tests; synchronises): the persistent scan's blocks meet at step counters with a
    BOUNDED spin; if the partners of a block were never scheduled (more persistent kernels resident than the GPU has
    CUs, e.g. several processes sharing one device) the scan writes an error word instead of hanging -- read it."""
    if os.environ.get("FOCR_LSTM_CHECK", "0") != "1":
        return
    ngroups = (batch + 31) // 32 * 2
    if _lib.load().focr_get_tuning(2) == 0 or _lib.get_precision() == 0 or 8 * ngroups > 256:
        return                                           # per-step launches: the flag words are not used
    # rnn.hip LP_FLAG_BYTES at the end of the workspace, or the call site's persistent flag block (_lstm_flags)
    flags = ws if ws.dtype == torch.int32 else ws[-1024:].view(torch.int32)
    if int(flags[ngroups].item()) != 0:
        raise RuntimeError("persistent LSTM scan: a step-counter wait timed out (partner blocks not resident)")


# Frozen recurrent weights (the recognizer of the training step): their bf16 hi / lo split -- and, for the backward scan,
# the transposed split -- is the same in every step.  Prepared once per (tensor, version) and handed to the scans
# (focr_lstm_bidir_*_pw), which then skip their per-call split launch (4 launches + 4 dependent gaps per step).  A call
# with TRAINABLE weights drops the entry: raw-pointer optimisers change them behind autograd's version counter.
_LSTM_SPLIT = {}      # id(whh) -> (weakref, version, data_ptr, {backward flag: uint8 tensor})


def _lstm_prepared(whh, backward):
    if whh.requires_grad or not whh.is_cuda or whh.shape[-1] != 256 or _lib.get_precision() == 0:
        _LSTM_SPLIT.pop(id(whh), None)
        return None
    e = _LSTM_SPLIT.get(id(whh))
    if e is None or e[0]() is not whh or e[1] != whh._version or e[2] != whh.data_ptr():
        if len(_LSTM_SPLIT) > 64:
            for k_ in [k_ for k_, v_ in _LSTM_SPLIT.items() if v_[0]() is None]:
                del _LSTM_SPLIT[k_]
        e = (weakref.ref(whh), whh._version, whh.data_ptr(), {})
        _LSTM_SPLIT[id(whh)] = e
    buf = e[3].get(backward)
    if buf is None:
        buf = torch.empty(_lib.load().focr_lstm_split_bytes(256), device=whh.device, dtype=torch.uint8)
        _lib.call("focr_lstm_prepare_weights", _p(whh), ctypes.c_void_p(buf.data_ptr()), 256, int(backward), _stream())
        e[3][backward] = buf
    return buf


def _lstm_flags(whh, backward, batch, t_len):
    """(flags tensor, base) of the persistent scan at this call site -- (frozen weights, pass, batch, T) -- or (None, 0).
    The step counters are zeroed ONCE and keep counting: a scan adds 8 (T - 1) to each of its groups' words, so the next
    call's targets start at the running sum (mod 2^32) and no memset launch precedes the scan.  Calls of one site are
    ordered by the stream; two streams scanning with the SAME weights at the same time would share the words -- the sites
    of this library (one recognizer per engine step) never do."""
    e = _LSTM_SPLIT.get(id(whh))
    if e is None or e[0]() is not whh or not _lib.load().focr_lstm_persistent_usable(int(batch), 256):
        return None, 0                      # (per-step launches do not touch the counters: the base must not advance)
    if torch.cuda.is_current_stream_capturing():
        # a recorded step (engine.TrainStep replay) re-issues this launch with the SAME scalar arguments every time: the
        # counters must start from zero in every replay, i.e. the workspace flag block + its memset (both recorded)
        return None, 0
    key = ("flags", int(backward), int(batch), int(t_len), torch.cuda.current_stream().cuda_stream)
    st = e[3].get(key)
    if st is None:
        # [flag words, base, calls, (pinned error copy, event) of the last health check]
        st = [torch.zeros(256, device=whh.device, dtype=torch.int32), 0, 0, None]
        e[3][key] = st
    # Health check without a synchronisation: every 16th call copies the sticky error word (set by a bounded
    # wait that timed out: partner blocks not resident) to pinned memory; the NEXT check reads the copy if it has
    # landed.  After a timeout the counters no longer match `base` and every later scan of this site would time out too:
    # the flag block is zeroed and the base restarts at 0, and the caller is told once.
    st[2] += 1
    if st[3] is not None and st[3][1].query():
        if int(st[3][0][0]) != 0:
            import warnings
            warnings.warn("persistent LSTM scan: a step-counter wait timed out (partner blocks not resident -- several "
                          "processes on one device?); the affected step's recognizer output was NaN-poisoned, the counters "
                          "are reset.  focr_set_tuning(2, 0) selects per-step launches.")
            st[0].zero_()
            st[1] = 0
        st[3] = None
    if st[3] is None and st[2] % 16 == 0:
        ngroups = (int(batch) + 31) // 32 * 2
        pin = torch.empty(1, dtype=torch.int32, pin_memory=True)
        pin.copy_(st[0][ngroups:ngroups + 1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st[3] = (pin, ev)
    return st[0], st


def _lstm_advance(st, t_len):
    """the scan was launched: its groups' words end 8 (T - 1) higher"""
    st[1] = (st[1] + 8 * (t_len - 1)) & 0xFFFFFFFF


class _LSTMRecur(torch.autograd.Function):
    """gx: [rows, 2*4H] with row(t,b) = t*st_t + b*st_b;  returns hseq [T,B,2H].
    Gradients: gx always; W_hh / b_hh when they require one (trainable recognizer):
    dW_hh[d] = sum_t dgates_t[d]^T h_{t -/+ 1}[d] as ONE strided GEMM per direction over the (T-1) B row pairs,
    db_hh = column sums of the gate gradients."""

    @staticmethod
    def forward(ctx, gx, whh, bhh, t_len, batch, st_t, st_b):
        hid = whh.shape[-1]
        _chk(gx, whh, bhh)
        hseq = torch.empty((t_len, batch, 2 * hid), device=gx.device)
        gates = torch.empty((t_len, batch, 2, 4 * hid), device=gx.device)
        cseq = torch.empty((t_len, batch, 2, hid), device=gx.device)
        ws = torch.empty(_lib.load().focr_lstm_ws_bytes(t_len, batch, hid, 0), device=gx.device, dtype=torch.uint8)
        if ctx.needs_input_grad[0]:
            current_context().prefetch_masks_early()
        wsp = _lstm_prepared(whh, 0)
        fl, fst = _lstm_flags(whh, 0, batch, t_len) if wsp is not None else (None, 0)
        _lib.call("focr_lstm_bidir_fwd_pw", _p(gx), _p(whh), _p(bhh), _p(hseq), _p(gates), _p(cseq), _p(ws),
                  ctypes.c_void_p(wsp.data_ptr()) if wsp is not None else _NULL,
                  ctypes.c_void_p(fl.data_ptr()) if fl is not None else _NULL, fst[1] if fl is not None else 0, t_len,
                  batch, hid, st_t, st_b, _stream())
        if fl is not None:
            _lstm_advance(fst, t_len)                      # only after the launch succeeded (a raise leaves base alone)
        _lstm_check(ws if fl is None else fl, batch)
        ctx.cfg = (t_len, batch, hid, st_t, st_b, tuple(gx.shape))
        ctx.save_for_backward(whh, gates, cseq, hseq if (whh.requires_grad or bhh.requires_grad) else None)
        return hseq

    @staticmethod
    def backward(ctx, dh):
        whh, gates, cseq, hseq = ctx.saved_tensors
        t_len, batch, hid, st_t, st_b, gshape = ctx.cfg
        dh = dh.contiguous()
        dgx = torch.empty(gshape, device=dh.device)
        carry = torch.empty((2, batch, hid), device=dh.device)
        ws = torch.empty(_lib.load().focr_lstm_ws_bytes(t_len, batch, hid, 1), device=dh.device, dtype=torch.uint8)
        wsp = _lstm_prepared(whh, 1)
        fl, fst = _lstm_flags(whh, 1, batch, t_len) if wsp is not None else (None, 0)
        _lib.call("focr_lstm_bidir_bwd_pw", _p(dh), _p(whh), _p(gates), _p(cseq), _p(dgx), _p(carry), _p(ws),
                  ctypes.c_void_p(wsp.data_ptr()) if wsp is not None else _NULL,
                  ctypes.c_void_p(fl.data_ptr()) if fl is not None else _NULL, fst[1] if fl is not None else 0, t_len,
                  batch, hid, st_t, st_b, _stream())
        if fl is not None:
            _lstm_advance(fst, t_len)
        _lstm_check(ws if fl is None else fl, batch)
        dwhh = dbhh = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            rows = t_len * batch
            # gate gradients in the row order of hseq (t-major); the CRNN's first layer addresses them batch-major
            dg = dgx if (st_t == batch and st_b == 1) else \
                dgx.view(batch, t_len, 8 * hid).transpose(0, 1).contiguous().view(rows, 8 * hid)
            if ctx.needs_input_grad[2]:
                dbhh = torch.empty((2, 4 * hid), device=dh.device)
                _lib.call("focr_colsum", _p(dg), _p(dbhh), rows, 8 * hid, 8 * hid, _stream())
            if ctx.needs_input_grad[1]:
                dwhh = torch.zeros((2, 4 * hid, hid), device=dh.device)
                if t_len > 1:
                    m = (t_len - 1) * batch
                    nws = _lib.load().focr_conv2d_wgrad_ws_floats(m, 1, 1, hid, 4 * hid, 1, 1, 0, 0)
                    wsw = torch.empty(max(nws, 1), device=dh.device)
                    # forward direction: rows t = 1.. pair with h of t - 1;  reverse: rows t = ..T-2 with h of t + 1
                    _lib.call("focr_conv2d_wgrad", _po(hseq, 0), _po(dg, batch * 8 * hid), _po(dwhh, 0), _NULL, m, 1,
                              1, hid, 4 * hid, 1, 1, 0, 0, 8 * hid, 2 * hid, 1, _p(wsw), nws, _stream())
                    _lib.call("focr_conv2d_wgrad", _po(hseq, batch * 2 * hid + hid), _po(dg, 4 * hid),
                              _po(dwhh, 4 * hid * hid), _NULL, m, 1, 1, hid, 4 * hid, 1, 1, 0, 0, 8 * hid, 2 * hid, 1,
                              _p(wsw), nws, _stream())
        return dgx, dwhh, dbhh, None, None, None, None


def lstm_recurrence(gx, whh, bhh, t_len, batch, st_t, st_b):
    return _LSTMRecur.apply(gx, whh, bhh, t_len, batch, st_t, st_b)


# ----------------------------------------------------------------------------------------
# bidirectional GRU of the TSRN blocks (recurrent part; the input projection is a `linear`)
# ----------------------------------------------------------------------------------------
_GRU_WHH_CROSS = os.environ.get("FOCR_GRU_WHH_CROSS", "1") != "0"


class _GRURecur(torch.autograd.Function):
    """gx [rows,192] (= x W_ih^T + b_ih, both directions), whh [2,96,32], bhh [2,96] -> h [rows,64].
    Sequences are addressed in place on the NHWC map: row(n,t) = (n//IC)*OS + (n%IC)*IS + t*TS."""

    @staticmethod
    def forward(ctx, gx, whh, bhh, nseq, t_len, ic, os_, is_, ts):
        _chk(gx, whh, bhh)
        rows = gx.shape[0]
        assert nseq * t_len == rows and whh.shape == (2, 96, 32)
        hseq = torch.empty((rows, 64), device=gx.device)
        gates = torch.empty((rows, 2, 128), device=gx.device)
        _lib.call("focr_gru_bidir_fwd", _p(gx), _p(whh), _p(bhh), _p(hseq), _p(gates), nseq, t_len, ic, os_,
                  is_, ts, _stream())
        ctx.cfg = (rows, nseq, t_len, ic, os_, is_, ts)
        ctx.targets = (_target(whh), _target(bhh))
        ctx.save_for_backward(whh, gates, hseq)
        return hseq

    @staticmethod
    def backward(ctx, dh):
        whh, gates, hseq = ctx.saved_tensors
        rows, nseq, t_len, ic, os_, is_, ts = ctx.cfg
        dh = dh.contiguous()
        dgx = torch.empty((rows, 192), device=dh.device)
        dgh = torch.empty((rows, 192), device=dh.device)
        hprev = torch.empty((rows, 2, 32), device=dh.device)
        _lib.call("focr_gru_bidir_bwd", _p(dh), _p(whh), _p(gates), _p(hseq), _p(dgx), _p(dgh), _p(hprev), nseq,
                  t_len, ic, os_, is_, ts, _stream())
        tw, tb = ctx.targets          # slices of the engine's flat gradient buffer (zeroed once per step): accumulate
        pz = int(tw is not None and tb is not None)
        dwhh = tw if pz else torch.empty((2, 96, 32), device=dh.device)
        dbhh = tb if pz else torch.empty((2, 96), device=dh.device)
        nws = _lib.load().focr_conv2d_wgrad_ws_floats(rows, 1, 1, 64, 192, 1, 1, 0, 0)
        if nws > 0 and _lib.get_precision() != 0 and rows >= 4096 and rows % 16 == 0 and _GRU_WHH_CROSS:
            # both directions from ONE streaming pass over dgh [rows, 192] and h_prev [rows, 64]: the [192 x 64] cross
            # product (its diagonal blocks are the two dW_hh, focr_gru_whh_extract) + the column sums of dgh = db_hh of both
            # directions.  The two strided 96 x 32 launches on the generic fp32 kernel it replaces read the same rows twice
            # at 73 us each (profiles/r06c_c1_bygrid.txt).
            cross = torch.empty(192 * 64 + 192, device=dh.device)          # matrix, then the 192 bias sums (overwritten)
            wsw = torch.empty(nws, device=dh.device)
            _lib.call("focr_conv2d_wgrad", _p(hprev), _p(dgh), _p(cross), _po(cross, 192 * 64), rows, 1, 1, 64, 192, 1, 1,
                      0, 0, 192, 64, 0, _p(wsw), nws, _stream())
            _lib.call("focr_gru_whh_extract", _p(cross), _p(dwhh), _p(dbhh), pz, _stream())
        else:
            for d in (0, 1):     # dW_hh[d] = dgh[:, d]^T hprev[:, d]  -- the generic wgrad on strided views
                _lib.call("focr_conv2d_wgrad", _po(hprev, 32 * d), _po(dgh, 96 * d), _po(dwhh, 96 * 32 * d),
                          _po(dbhh, 96 * d), rows, 1, 1, 32, 96, 1, 1, 0, 0, 192, 64, pz, _NULL, 0, _stream())
        if pz:
            dwhh = dbhh = None
        return dgx, dwhh, dbhh, None, None, None, None, None, None


def gru_recurrence(gx, whh, bhh, nseq, t_len, ic, os_, is_, ts):
    return _GRURecur.apply(gx, whh, bhh, nseq, t_len, ic, os_, is_, ts)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        y = torch.empty_like(a)
        _lib.call("focr_axpy", _p(a), _p(b), _p(y), a.numel(), 1.0, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


# ----------------------------------------------------------------------------------------
# optimiser tail
# ----------------------------------------------------------------------------------------
def sumsq_workspace(device):
    """workspace of focr_grad_sumsq: element 0 receives the squared gradient norm"""
    return torch.zeros(_lib.load().focr_grad_sumsq_ws_floats(), device=device)


def grad_sumsq(flat_grad, out, gscale=1.0):
    _lib.call("focr_grad_sumsq", _p(flat_grad), _p(out), flat_grad.numel(), float(gscale), _stream())


def clip_adam(p, g, m, v, sumsq, lr, beta1, beta2, eps, step, max_norm, gscale=1.0):
    _lib.call("focr_clip_adam", _p(p), _p(g), _p(m), _p(v), _p(sumsq), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), int(step), float(max_norm), float(gscale), _stream())


def clip_adam_state(p, g, m, v, sumsq, lr, beta1, beta2, eps, state, max_norm, gscale=1.0):
    """clip + Adam with the step count / bias corrections read from the device-resident step state (focr_step_advance)"""
    _lib.call("focr_clip_adam_state", _p(p), _p(g), _p(m), _p(v), _p(sumsq), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), _p(state), float(max_norm), float(gscale), _stream())


class StepState:
    """64 bytes of device memory an engine owns (csrc/focr_core.hip): dropout epoch, optimiser step count and the Adam
    bias corrections of that step.  Advanced by ONE tiny launch at the start of every step, read by the kernels -- so
    the host passes no per-step scalar, and a recorded launch sequence (replay.py) is a correct next step."""

    def __init__(self, device, beta1, beta2):
        self.buf = torch.zeros(_lib.load().focr_step_state_bytes() // 8, device=device, dtype=torch.int64)
        self.betas = (float(beta1), float(beta2))

    def advance(self):
        _lib.call("focr_step_advance", _p(self.buf), self.betas[0], self.betas[1], _stream())

    def set_counts(self, epoch, t):
        """(checkpoint resume) the next advance() makes these epoch + 1 / t + 1"""
        self.buf[:2].copy_(torch.tensor([int(epoch), int(t)], dtype=torch.int64))

    def counts(self):
        e, t = self.buf[:2].tolist()        # synchronises
        return int(e), int(t)

    def bind(self):
        _lib.call("focr_set_seed_epoch", _p(self.buf))

    @staticmethod
    def unbind():
        _lib.call("focr_set_seed_epoch", None)

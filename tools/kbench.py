#!/usr/bin/env python3
"""Kernel micro-benchmarks (on-stream event timing, interleaved rounds) for A/B work.
usage: [FOCR_LIB=path/to/variant.so] python tools/kbench.py [attn] [conv] [lstm] [--batch 128]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fudanocr_amd import _lib, kernels as K   # noqa: E402

B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 128
what = [a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()] or ["attn", "conv"]



def timeit(fn, iters=8, warm=2):
    for _ in range(warm):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3      # median, min in us


if "--fp32" in sys.argv:
    _lib.set_precision(0)
if "--prec" in sys.argv:
    _lib.set_precision(int(sys.argv[sys.argv.index("--prec") + 1]))
print("lib:", _lib.LIB_PATH, " batch", B, " precision", _lib.get_precision())
g = torch.Generator(device="cuda").manual_seed(0)
if "attn" in what:
    q, k, v = (torch.randn(B, 1024, 128, device="cuda", generator=g) for _ in range(3))
    do = torch.randn(B, 1024, 128, device="cuda", generator=g)
    fl = 4.0 * B * 4 * 1024 * 1024 * 32
    for p in (0.1, 0.0):
        o = K._Attention.apply(q, k, v, 4, p, 1234)
        lse = torch.empty(B, 4, 1024, device="cuda")
        mask = torch.empty(B, 4, 32, 32, 32, device="cuda", dtype=torch.int32)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        work = torch.empty(B, 4, 1024, device="cuda")

        def fwd():
            _lib.call("focr_attention_fwd", K._p(q), K._p(k), K._p(v), K._p(o), K._p(lse), K._p(mask), B, 4, 1024, 128, 128,
                      1 / math.sqrt(32), p, 1234, K._stream())

        def bwd():
            _lib.call("focr_attention_bwd", K._p(q), K._p(k), K._p(v), K._p(o), K._p(do), K._p(lse), K._p(mask),
                      K._p(dq), K._p(dk), K._p(dv), K._p(work), B, 4, 1024, 128, 128, 1 / math.sqrt(32), p, K._stream())
        m, mn = timeit(fwd)
        print("attn fwd  p=%.1f  median %8.1f us  min %8.1f us  %6.1f TF (algorithmic)" % (p, m, mn, fl / mn / 1e6))
        m, mn = timeit(bwd)
        print("attn bwd  p=%.1f  median %8.1f us  min %8.1f us  %6.1f TF (algorithmic 2.5x fwd)" % (p, m, mn, 2.5 * fl / mn / 1e6))
if "conv" in what:
    shapes = [("srb 3x3 64->64", (B, 16, 64, 64), 64, 3, 1), ("linear 128->128", (B * 1024, 1, 1, 128), 128, 1, 0),
              ("linear 128->64", (B * 1024, 1, 1, 128), 64, 1, 0), ("up 3x3 64->256", (B, 16, 64, 64), 256, 3, 1),
              ("crnn 3x3 256->512 4x26", (B, 4, 26, 256), 512, 3, 1), ("crnn 3x3 128->256 8x25", (B, 8, 25, 128), 256, 3, 1),
              ("crnn 3x3 64->128 16x50", (B, 16, 50, 64), 128, 3, 1)]
    for name, xs, cout, ks, pad in shapes:
        n, h, w, cin = xs
        x = torch.randn(xs, device="cuda", generator=g)
        wt = torch.randn(cout, ks, ks, cin, device="cuda", generator=g)
        bias = torch.randn(cout, device="cuda", generator=g)
        y = torch.empty(n, h, w, cout, device="cuda")
        dw = torch.empty_like(wt)
        db = torch.empty(cout, device="cuda")
        fl = 2.0 * n * h * w * cout * ks * ks * cin
        nws = _lib.load().focr_conv2d_wgrad_ws_floats(n, h, w, cin, cout, ks, ks, pad, pad)
        wsw = torch.empty(nws, device="cuda") if nws > 0 else None

        def fwd():
            _lib.call("focr_conv2d_fwd", K._p(x), K._p(wt), K._p(bias), K._NULL, K._p(y), n, h, w, cin, cout, ks, ks,
                      pad, pad, 1.0, 0, 0, 0, 0, K._stream())

        def wg():
            _lib.call("focr_conv2d_wgrad", K._p(x), K._p(y), K._p(dw), K._p(db), n, h, w, cin, cout, ks, ks, pad, pad,
                      0, 0, 0, K._p(wsw), nws, K._stream())
        m, mn = timeit(fwd)
        m2, mn2 = timeit(wg)
        print("%-26s fwd median %7.1f us min %7.1f us %6.1f TF | wgrad median %7.1f us min %7.1f %6.1f TF"
              % (name, m, mn, fl / mn / 1e6, m2, mn2, fl / mn2 / 1e6))
if "lstm" in what:
    t, hid = 26, 256
    gx = torch.randn(t * B, 2048, device="cuda", generator=g)
    whh = torch.randn(2, 1024, 256, device="cuda", generator=g) * 0.05
    bhh = torch.zeros(2, 1024, device="cuda")
    hseq = torch.empty(t, B, 512, device="cuda")
    gates = torch.empty(t, B, 2, 1024, device="cuda")
    cseq = torch.empty(t, B, 2, 256, device="cuda")
    dgx = torch.empty_like(gx)
    carry = torch.empty(2, B, 256, device="cuda")
    wsf = torch.empty(_lib.load().focr_lstm_ws_bytes(t, B, hid, 0), device="cuda", dtype=torch.uint8)
    wsb = torch.empty(_lib.load().focr_lstm_ws_bytes(t, B, hid, 1), device="cuda", dtype=torch.uint8)

    def fwd():
        _lib.call("focr_lstm_bidir_fwd", K._p(gx), K._p(whh), K._p(bhh), K._p(hseq), K._p(gates), K._p(cseq), K._p(wsf), t, B,
                  hid, B, 1, K._stream())

    def bwd():
        _lib.call("focr_lstm_bidir_bwd", K._p(hseq), K._p(whh), K._p(gates), K._p(cseq), K._p(dgx), K._p(carry), K._p(wsb), t, B,
                  hid, B, 1, K._stream())
    print("lstm fwd (26 steps) median %.1f us" % timeit(fwd)[0], " bwd median %.1f us" % timeit(bwd)[0])

if "out9" in what:
    for (h, w) in ((32, 128), (16, 64)):
        x = torch.randn(B, h, w, 64, device="cuda", generator=g)
        wt = torch.randn(3, 9, 9, 64, device="cuda", generator=g)
        bias = torch.randn(3, device="cuda", generator=g)
        y = torch.empty(B, h, w, 3, device="cuda")

        def fwd():
            _lib.call("focr_conv9x9_small_cout_fwd", K._p(x), K._p(wt), K._p(bias), K._p(y), B, h, w, 64, 3, K._stream())
        m, mn = timeit(fwd)
        print("conv9x9 64->3 %dx%d fwd median %8.1f us  min %8.1f us" % (h, w, m, mn))
if "ln" in what:
    rows = B * 1024
    x = torch.randn(rows, 128, device="cuda", generator=g)
    r = torch.randn(rows, 128, device="cuda", generator=g)
    a, b2 = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    y, dx = torch.empty_like(x), torch.empty_like(x)
    mean, rinv = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    da, db = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")

    def fwd():
        _lib.call("focr_layernorm_fwd", K._p(x), K._p(r), K._p(a), K._p(b2), K._p(y), K._p(mean), K._p(rinv), rows, 128,
                  1e-6, K._stream())

    def bwd():
        _lib.call("focr_layernorm_bwd", K._p(y), K._p(x), K._p(r), K._p(a), K._p(mean), K._p(rinv), K._p(dx), K._p(da),
                  K._p(db), rows, 128, 1e-6, 1, K._stream())
    m, mn = timeit(fwd)
    print("layernorm fwd median %8.1f us  min %8.1f us" % (m, mn))
    m, mn = timeit(bwd)
    print("layernorm bwd median %8.1f us  min %8.1f us" % (m, mn))
if "gru" in what:
    # TSRN GruBlock scans at B = 128 on the 16 x 64 map: gru1 vertical (nseq = B*64, T = 16), gru2 horizontal (B*16, 64)
    rows = B * 16 * 64
    gxg = torch.randn(rows, 192, device="cuda", generator=g)
    whh = torch.randn(2, 96, 32, device="cuda", generator=g) * 0.2
    bhh = torch.randn(2, 96, device="cuda", generator=g) * 0.1
    hs = torch.empty(rows, 64, device="cuda")
    gts = torch.empty(rows, 2, 128, device="cuda")
    dh = torch.randn(rows, 64, device="cuda", generator=g)
    dgx, dgh = torch.empty(rows, 192, device="cuda"), torch.empty(rows, 192, device="cuda")
    hpv = torch.empty(rows, 2, 32, device="cuda")
    for name, (nseq, t, ic, os_, is_, ts) in (("gru1 vertical T=16", (B * 64, 16, 64, 16 * 64, 1, 64)),
                                              ("gru2 horizontal T=64", (B * 16, 64, 1, 64, 0, 1))):
        def fwd():
            _lib.call("focr_gru_bidir_fwd", K._p(gxg), K._p(whh), K._p(bhh), K._p(hs), K._p(gts), nseq, t, ic, os_, is_, ts,
                      K._stream())

        def bwd():
            _lib.call("focr_gru_bidir_bwd", K._p(dh), K._p(whh), K._p(gts), K._p(hs), K._p(dgx), K._p(dgh), K._p(hpv), nseq,
                      t, ic, os_, is_, ts, K._stream())
        m, mn = timeit(fwd)
        m2, mn2 = timeit(bwd)
        print("%-22s fwd median %8.1f us (%.2f us/step)  bwd median %8.1f us (%.2f us/step)" % (name, m, m / t, m2, m2 / t))

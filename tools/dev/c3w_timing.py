#!/usr/bin/env python3
"""Phase timing of conv3x3_c64_wgrad_kernel from in-kernel cycle stamps (library built with -DC3W_TIMING):
FOCR_LIB=.../libfocr_hip_c3w_timing.so python tools/dev/c3w_timing.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fudanocr_amd import _lib, kernels as K   # noqa: E402

B = 128
n, h, w, cin, cout = B, 16, 64, 64, 64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, h, w, cin, device="cuda", generator=g)
dy = torch.randn(n, h, w, cout, device="cuda", generator=g)
dw = torch.zeros(cout, 3, 3, cin, device="cuda")
db = torch.zeros(cout, device="cuda")
nws = _lib.load().focr_conv2d_wgrad_ws_floats(n, h, w, cin, cout, 3, 3, 1, 1)
ws = torch.empty(nws, device="cuda")
for _ in range(5):
    _lib.call("focr_conv2d_wgrad", K._p(x), K._p(dy), K._p(dw), K._p(db), n, h, w, cin, cout, 3, 3, 1, 1, 0, 0, 1, K._p(ws),
              nws, K._stream())
torch.cuda.synchronize()
out = np.zeros((1024, 16), dtype=np.uint64)
lib = _lib.load()
lib.focr_debug_c3w_stamps.argtypes = [ctypes.c_void_p]
assert lib.focr_debug_c3w_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
st = out[:256, :8].astype(np.int64)
names = ["start->primed (prologue: loads, split, staging, barrier)", "first step: barrier->products issued",
         "first step: products issued->trailing barrier", "trailing barrier->second step staged",
         "second step staged->its barrier", "second barrier->loop end (6 more steps)", "loop end->partials stored"]
pairs = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)]
print("cycles of s_memtime (100 MHz constant clock on gfx9: 1 tick = 10 ns), median / min / max over 256 blocks")
for nm, (a, b) in zip(names, pairs):
    d = st[:, b] - st[:, a]
    print("%-62s %8.0f %8d %8d" % (nm, np.median(d), d.min(), d.max()))
tot = st[:, 7] - st[:, 0]
print("%-62s %8.0f %8d %8d" % ("block total", np.median(tot), tot.min(), tot.max()))
print("first block start -> last block end: %d ticks" % (st[:, 7].max() - st[:, 0].min()))

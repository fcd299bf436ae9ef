"""conv3x3_halo_kernel at the multi-slice shapes of the SLD / text-focus ResNets (Cin >= 128), on-stream event timing, both operand-plane
counts.  FOCR_LIB selects a library variant; FOCR_H3_GROUP_FAST=0/1 the block order (A/B in one process is not possible: read once)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K
_lib.load(); _lib.set_precision(3)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


SHAPES = ((16, 8, 32, 512, 512), (16, 8, 32, 256, 256), (16, 8, 32, 512, 1024), (16, 16, 64, 128, 128), (8, 16, 16, 512, 512),
          (128, 8, 32, 512, 512), (128, 8, 32, 256, 256), (128, 8, 32, 512, 1024), (128, 16, 64, 128, 128), (128, 16, 64, 64, 128),
          (32, 16, 16, 512, 512), (32, 16, 16, 256, 256), (32, 8, 8, 1024, 1024), (128, 16, 64, 64, 64))
for (n, h, w, cin, cout) in SHAPES:
    x = torch.randn(n, h, w, cin, device=dev, generator=g)
    wt = torch.randn(cout, 3, 3, cin, device=dev, generator=g) * (1.0 / (9 * cin) ** 0.5)
    y = torch.empty(n, h, w, cout, device=dev)
    frag = torch.empty(_lib.load().focr_weight_frag_bytes(cout, 9 * cin), device=dev, dtype=torch.uint8)
    _lib.call("focr_weight_prep_frag", K._p(wt), ctypes.c_void_p(frag.data_ptr()), cout, 3, 3, cin, 0, K._stream())
    line = "%4d x %2d x %2d  %4d -> %4d " % (n, h, w, cin, cout)
    fl = 2.0 * n * h * w * 9 * cin * cout
    for planes in (2, 1):
        f = lambda: _lib.call("focr_conv3x3_frag_fwd", K._p(x), ctypes.c_void_p(frag.data_ptr()), K._NULL, K._NULL, K._p(y), K._NULL,
                              n, h, w, cin, cout, 1.0, 0, planes, 0, 0, 0, K._stream())
        med, mn = timeit(f)
        line += " | planes %d: median %7.1f min %7.1f us %6.1f TFLOP/s alg" % (planes, med, mn, fl / mn / 1e6)
    if n * h * w * cin <= 1 << 22:
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.permute(0, 3, 1, 2).double(), padding=1).permute(0, 2, 3, 1)
        line += "  err(planes 1) %.1e" % float((y.double() - ref).abs().max() / ref.abs().max())
    print(line)

"""conv_wgrad_bx3_wide_kernel at the SLD encoder's shapes (B = 32): 3x3 weight gradients of 256 / 512 / 1024-channel layers on
16 x 16 maps, on-stream event timing.  FOCR_LIB selects a library variant (ablations: -DWGW_ABL=bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K
_lib.load(); _lib.set_precision(3)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


for (n, h, w, cin, cout) in ((32, 16, 16, 256, 256), (32, 16, 16, 512, 512), (32, 16, 16, 512, 1024), (32, 16, 16, 1024, 1024)):
    x = torch.randn(n, h, w, cin, device=dev, generator=g)
    dy = torch.randn(n, h, w, cout, device=dev, generator=g)
    dw = torch.zeros(cout, 3, 3, cin, device=dev)
    db = torch.zeros(cout, device=dev)
    m = n * h * w
    nws = _lib.load().focr_conv2d_wgrad_ws_floats(n, h, w, cin, cout, 3, 3, 1, 1)
    ws = torch.empty(max(nws, 1), device=dev)
    f = lambda: _lib.call("focr_conv2d_wgrad", K._p(x), K._p(dy), K._p(dw), K._p(db), n, h, w, cin, cout, 3, 3, 1, 1, cout, cin, 1,
                          K._p(ws), nws, K._stream())
    med, mn = timeit(f)
    fl = 2.0 * m * 9 * cin * cout
    err = ""
    if cin == 256:                      # correctness against torch (fp64 on the host is too slow: fp32 conv backward on the device)
        dw.zero_(); db.zero_(); f(); torch.cuda.synchronize()
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).double().cpu(), (cout, cin, 3, 3), dy.permute(0, 3, 1, 2).double().cpu(), padding=1)
        got = dw.permute(0, 3, 1, 2).double().cpu()
        err = "  rel-to-max err %.2e, bias err %.2e" % (float((got - ref).abs().max() / ref.abs().max()),
                                                       float((db.double().cpu() - dy.double().sum((0, 1, 2)).cpu()).abs().max() / dy.double().sum((0, 1, 2)).abs().max().cpu()))
    print("%4d -> %4d  median %7.1f min %7.1f us  %6.1f TFLOP/s algorithmic%s" % (cin, cout, med, mn, fl / mn / 1e6, err))

"""large 1 x 1 layers (plain GEMMs): generic implicit-GEMM kernel (FOCR_GEMM_BIG=0) vs the 256 x 128 tile kernel (gemm_big.hip)"""
import os, sys
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


for (m, k, n) in ((32768, 1024, 1024), (8192, 1024, 1024), (32768, 512, 1024), (16384, 1024, 2048), (131072, 256, 256)):
    x = torch.randn(m, 1, 1, k, device=dev, generator=g)
    w = torch.randn(n, 1, 1, k, device=dev, generator=g) * (1.0 / k ** 0.5)
    y = torch.empty(m, 1, 1, n, device=dev)
    f = lambda: _lib.call("focr_conv2d_fwd", K._p(x), K._p(w), K._NULL, K._NULL, K._p(y), m, 1, 1, k, n, 1, 1, 0, 0, 1.0, 0, 0, 0, 0, K._stream())
    med, mn = timeit(f)
    err = ""
    if m <= 8192:
        ref = x.view(m, k).double() @ w.view(n, k).double().t()
        err = "  rel-to-max err %.1e" % float((y.view(m, n).double() - ref).abs().max() / ref.abs().max())
    print("M %6d K %4d N %4d  median %7.1f min %7.1f us  %6.1f TFLOP/s alg%s" % (m, k, n, med, mn, 2.0 * m * k * n / mn / 1e6, err))

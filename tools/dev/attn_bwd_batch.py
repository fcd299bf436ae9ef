"""attention backward (TBSRN shape: 4 heads x 32, 1024 tokens, dropout 0.1) against the batch: the single-pass kernel launches one block
per (batch, head) -- 64 blocks at the reference README's batch 16 -- the two-pass kernels one per 128 queries.  Tuning key 3 selects."""
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
    return ts[len(ts) // 2] * 1e3


heads, t, d, p = 4, 1024, 128, 0.1
for b in (4, 8, 16, 24, 32, 48, 64, 96, 128):
    qkv = torch.randn(b, t, 3 * d, device=dev, generator=g)
    o = torch.empty(b, t, d, device=dev); lse = torch.empty(b, heads, t, device=dev)
    mask = torch.empty((b, heads, t // 32, t // 32, 32), device=dev, dtype=torch.int32)
    _lib.call("focr_attention_dropout_mask", K._p(mask), b, heads, t, p, 1234, K._stream())
    scale = 1.0 / (d // heads) ** 0.5
    _lib.call("focr_attention_fwd_premasked", K._po(qkv, 0), K._po(qkv, d), K._po(qkv, 2 * d), K._p(o), K._p(lse), K._p(mask), b, heads, t,
              3 * d, d, scale, p, K._stream())
    do = torch.randn(b, t, d, device=dev, generator=g)
    dqkv = torch.empty_like(qkv); work = torch.empty(b, heads, t, device=dev)
    res = {}
    outs = {}
    for var in (2, 1, 0):
        _lib.call("focr_set_tuning", 3, var)
        f = lambda: _lib.call("focr_attention_bwd", K._po(qkv, 0), K._po(qkv, d), K._po(qkv, 2 * d), K._p(o), K._p(do), K._p(lse), K._p(mask),
                              K._po(dqkv, 0), K._po(dqkv, d), K._po(dqkv, 2 * d), K._p(work), b, heads, t, 3 * d, d, scale, p, K._stream())
        res[var] = timeit(f)
        f(); torch.cuda.synchronize(); outs[var] = dqkv.clone()
    _lib.call("focr_set_tuning", 3, 2)
    err = float((outs[2] - outs[1]).abs().max() / outs[1].abs().max())
    print("B = %3d  single pass %7.1f us   two passes (dq2) %7.1f us   two passes (dq) %7.1f us   max diff single vs two %.1e" % (b, res[2], res[1], res[0], err))

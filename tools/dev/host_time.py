"""Host-side cost of the c3 training step without a profiler: wall time of enqueueing a step at a small batch (the GPU never
back-pressures the host), and the number of autograd nodes / library calls per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib
_lib.load(); _lib.set_precision(3)
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
s = TrainStep(net, crit, dropout=True)
lr, hr, labels = make_batch(8, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
for _ in range(10): s(lr, hr, encoded=enc)
torch.cuda.synchronize()
ts = []
for _ in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter(); s(lr, hr, encoded=enc); ts.append(time.perf_counter() - t0)
ts.sort()
print("host enqueue per step: median %.2f ms, min %.2f ms" % (ts[len(ts) // 2] * 1e3, ts[0] * 1e3))
# autograd nodes of one step's graph
with torch.enable_grad():
    from fudanocr_amd import kernels as K
    with K.use_context(s.ctx):
        s._set_modes(); s.ctx.frags = s.frags; s.frags.refresh()
        sr = s.model(lr); loss = s.crit(sr, hr, None, enc)[0]
    seen, stack = set(), [loss.grad_fn]
    while stack:
        f = stack.pop()
        if f is None or f in seen: continue
        seen.add(f); stack.extend(n for n, _ in f.next_functions)
    import collections
    names = collections.Counter(type(f).__name__ for f in seen)
    print("autograd nodes:", len(seen), dict(names.most_common(12)))
    s.ctx.frags = None
calls = _lib.call_count() if hasattr(_lib, "call_count") else None
print("library call counter:", calls)

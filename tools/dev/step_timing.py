"""host enqueue time vs. GPU time of the c3 training step (is the step launch-bound?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib
_lib.load(); _lib.set_precision(3)
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
for side, lw in ((True, 1), (True, 0), (False, 1), (False, 0)):
    _lib.call("focr_set_tuning", 0, lw)
    step_ = TrainStep(net, crit, dropout=True, wgrad_side_stream=side)
    lr, hr, labels = make_batch(batch, 1234)
    lr, hr = lr.to(dev), hr.to(dev)
    enc = crit.encode(labels, dev)
    for _ in range(10): step_(lr, hr, encoded=enc)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): step_(lr, hr, encoded=enc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("lwgrad_stream=%d side=%s batch=%d: host enqueue %.2f ms/step, total %.2f ms/step" % (lw, side, batch, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    # host-only cost: count C-ABI calls per step
    cnt = {"n": 0}
    orig = _lib.call
    def counting(name, *a):
        cnt["n"] += 1
        return orig(name, *a)
    _lib.call = counting
    import fudanocr_amd.kernels as K
    K._lib.call = counting
    step_(lr, hr, encoded=enc)
    torch.cuda.synchronize()
    _lib.call = orig; K._lib.call = orig
    print("  C-ABI calls per step:", cnt["n"])

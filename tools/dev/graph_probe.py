"""feasibility probe: can the whole c3 training step be captured in a hipGraph as it is (seeds / Adam step baked)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K
_lib.load(); _lib.set_precision(3)
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
s = TrainStep(net, crit, dropout=True)
s.ctx.mask_prefetch = False                      # probe: masks drawn inline (baked seeds)
lr, hr, labels = make_batch(batch, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(5): s(lr, hr, encoded=enc)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        out = s(lr, hr, encoded=enc)
    torch.cuda.synchronize()
    print("capture ok; loss", out["loss"].item())
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("replay: host %.3f ms/step, total %.2f ms/step, loss %.5f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, out["loss"].item()))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:600])

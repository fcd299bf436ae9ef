"""A/B timing of the c3 training step inside ONE process (variants interleaved, best of several rounds): host enqueue
time vs. total time per step.   usage: python tools/dev/step_timing.py [batch] [variant ...]
variants: base | nomaskpf (no keep-bit prefetch) | noside (no side stream) | nolw (generic linear wgrad)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib
_lib.load(); _lib.set_precision(int(os.environ.get("FOCR_PRECISION", "3")))
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = sys.argv[2:] or ["base", "nomaskpf"]
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
lr, hr, labels = make_batch(batch, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
steps = {}
for v in variants:
    steps[v] = TrainStep(net, crit, dropout=True, wgrad_side_stream=(v != "noside"))
    steps[v].ctx.mask_prefetch = v != "nomaskpf"

def apply(v):
    _lib.call("focr_set_tuning", 0, 0 if v == "nolw" else 1)
    for kv in v.split("+"):
        if kv.startswith("t") and "=" in kv:
            k, val = kv[1:].split("=")
            _lib.call("focr_set_tuning", int(k), int(val))

best = {v: (1e9, 1e9) for v in variants}
for rnd in range(4):
    for v in variants:
        apply(v)
        s = steps[v]
        for _ in range(4 if rnd else 10): s(lr, hr, encoded=enc)
        torch.cuda.synchronize()
        n = 15
        t0 = time.perf_counter()
        for _ in range(n): s(lr, hr, encoded=enc)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best[v] = (min(best[v][0], (t1 - t0) / n * 1e3), min(best[v][1], (t2 - t0) / n * 1e3))
for v in variants:
    print("%-12s batch=%d: host enqueue %.2f ms/step, total %.2f ms/step" % (v, batch, best[v][0], best[v][1]))

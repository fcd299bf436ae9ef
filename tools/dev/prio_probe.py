"""does running the step on a HIGH-priority stream (side stream stays low) shorten it?"""
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
lr, hr, labels = make_batch(128, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
steps = {"default": (TrainStep(net, crit, dropout=True), None),
         "main_high": (TrainStep(net, crit, dropout=True), torch.cuda.Stream(priority=-1))}
best = {k: 1e9 for k in steps}
for rnd in range(4):
    for name, (s, st) in steps.items():
        ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(5): s(lr, hr, encoded=enc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(15): s(lr, hr, encoded=enc)
            torch.cuda.synchronize()
            best[name] = min(best[name], (time.perf_counter() - t0) / 15 * 1e3)
for k, v in best.items(): print("%-10s %.2f ms/step" % (k, v))

"""ms/step in windows of 20 steps from a cold start (how long the box takes to reach its steady state)"""
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
crnn = (sys.argv[2] != "nocrnn") if len(sys.argv) > 2 else True
net, rec, crit = build_models(dev, "tbsrn", with_crnn=crnn)
s = TrainStep(net, crit, dropout=True)
lr, hr, labels = make_batch(batch, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev) if crnn else None
out = []
for w in range(15):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): s(lr, hr, encoded=enc)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 20 * 1e3)
print("batch %d crnn %s: " % (batch, crnn) + " ".join("%.2f" % v for v in out))

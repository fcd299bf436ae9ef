"""which Python lines issue the small torch copy / fill / elementwise kernels inside a training step"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from fudanocr_amd import _lib
_lib.load(); _lib.set_precision(3)
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
step = TrainStep(net, crit, dropout=True)
lr, hr, labels = make_batch(128, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
for _ in range(3): step(lr, hr, encoded=enc)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(lr, hr, encoded=enc)
    torch.cuda.synchronize()
from collections import Counter
c = Counter(ev.name[:70] for ev in prof.events())
for n, k in c.most_common(60):
    print(k, n)

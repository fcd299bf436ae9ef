"""step time of the c3 configuration with and without dropout (attention keep bits + FFN dropout): what the keep-bit
path costs inside the step (the attention kernels wait on scalar loads of mask words that are cold in-step)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib
_lib.load(); _lib.set_precision(3)
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
lr, hr, labels = make_batch(128, 1234)
lr, hr = lr.to(dev), hr.to(dev)
res = {}
for rep in range(2):
    for drop in (True, False):
        net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
        s = TrainStep(net, crit, dropout=drop)
        enc = crit.encode(labels, dev)
        for _ in range(15): s(lr, hr, encoded=enc)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): s(lr, hr, encoded=enc)
        torch.cuda.synchronize()
        res.setdefault(drop, []).append((time.perf_counter() - t0) / 40 * 1e3)
        del s, net, rec, crit
print("ms/step with dropout %s, without %s" % (["%.3f" % v for v in res[True]], ["%.3f" % v for v in res[False]]))

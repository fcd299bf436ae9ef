"""cProfile of the host side of the c3 training step (what the Python / ctypes / allocator overhead is made of)"""
import cProfile, pstats, os, sys, io
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
lr, hr, labels = make_batch(16, 1234)          # small batch: the GPU never back-pressures the host
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
for _ in range(10): s(lr, hr, encoded=enc)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): s(lr, hr, encoded=enc)
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
ps = pstats.Stats(pr, stream=st).sort_stats("tottime")
ps.print_stats(28)
print(st.getvalue()[:6000])

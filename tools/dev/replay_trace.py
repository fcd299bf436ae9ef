"""workload for rocprofv3 --kernel-trace: PHASE=eager|replay steps of the c3 step (same process set-up either way)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K, replay
_lib.load(); _lib.set_precision(int(os.environ.get("MODE", "3")))
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
dev = torch.device("cuda", 0)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
phase = os.environ.get("PHASE", "replay")
replay.N_LANES = int(os.environ.get("N_LANES", replay.N_LANES))
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
s = TrainStep(net, crit, dropout=True)
s.ctx.mask_prefetch = False
lr, hr, labels = make_batch(batch, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)
for _ in range(5): s(lr, hr, encoded=enc)
fn = lambda: s(lr, hr, encoded=enc)
if phase == "replay":
    r, out = replay.record(fn, lanes=[torch.cuda.current_stream(), s.ctx.side_stream_obj] if os.environ.get('OWN_LANES','1')=='1' else None)
    fn = r.launch
for _ in range(5): fn()
torch.cuda.synchronize()
n = 12
t0 = time.perf_counter()
for _ in range(n): fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s B=%d lanes=%d: host %.3f ms/step, total %.3f ms/step" % (phase, batch, replay.N_LANES, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))

"""feasibility + timing probe of the library's step replay (csrc/replay.hip): the whole c3 step captured once, then
re-issued from C.  Prints eager / replay / hipGraphLaunch step and host times and the replayed loss."""
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
net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
s = TrainStep(net, crit, dropout=True)
s.ctx.mask_prefetch = False                      # probe: masks drawn inline (baked seeds)
lr, hr, labels = make_batch(batch, 1234)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


for _ in range(5): s(lr, hr, encoded=enc)
h, t = timed(lambda: s(lr, hr, encoded=enc))
print("eager : host %.3f ms/step, total %.3f ms/step" % (h, t), flush=True)
try:
    r, out = replay.record(lambda: s(lr, hr, encoded=enc), lanes=[torch.cuda.current_stream(), s.ctx.side_stream_obj])
except Exception as e:
    print("record failed:", type(e).__name__, str(e)[:800]); sys.exit(1)
print("recorded:", r.info, flush=True)
r.launch(); torch.cuda.synchronize()
print("replayed loss %.6f" % out["loss"].item(), flush=True)
h, t = timed(r.launch)
print("replay: host %.3f ms/step, total %.3f ms/step, loss %.6f" % (h, t, out["loss"].item()), flush=True)
names, lanes = r.node_names(), r.node_lanes()
from collections import Counter
print("lanes:", Counter(lanes))
with open(os.path.join(os.environ.get("OUT", "."), "replay_nodes.txt"), "w") as f:
    for i, (n_, l_) in enumerate(zip(names, lanes)):
        f.write("%4d lane %d %s\n" % (i, l_, n_[:160]))
nk = r.probe("attn_bwd1")
for _ in range(3): r.launch()
print("probes on attn_bwd1:", nk, ["%.1f us" % (ms * 1e3) for _, ms, _ in r.probe_read()])
if os.environ.get("GRAPH_LAUNCH", "1") == "1":
    try:
        h, t = timed(r.graph.replay)
        print("hipGraphLaunch: host %.3f ms/step, total %.3f ms/step" % (h, t), flush=True)
    except Exception as e:
        print("hipGraphLaunch failed:", type(e).__name__, str(e)[:300])

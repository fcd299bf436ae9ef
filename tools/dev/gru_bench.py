"""GRU scans of the TSRN blocks at the c1 shapes (B = 128): vertical (8192 sequences x 16 steps) and horizontal (2048 x 64),
forward and backward, on-stream event timing.  FOCR_LIB selects a library variant (ablations: -DGRU_ABL=bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K
_lib.load(); _lib.set_precision(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
rows = B * 1024
gx = torch.randn(rows, 192, device=dev, generator=g)
whh = torch.randn(2, 96, 32, device=dev, generator=g) * 0.2
bhh = torch.randn(2, 96, device=dev, generator=g) * 0.1
dh = torch.randn(rows, 64, device=dev, generator=g)
hseq = torch.empty(rows, 64, device=dev); gates = torch.empty(rows, 2, 128, device=dev)
dgx = torch.empty(rows, 192, device=dev); dgh = torch.empty(rows, 192, device=dev); hprev = torch.empty(rows, 2, 32, device=dev)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


h, w = 16, 64
for tune in ([int(x) for x in os.environ["GRU_TUNE"].split(",")] if os.environ.get("GRU_TUNE") else [_lib.load().focr_get_tuning(5)]):
  _lib.call("focr_set_tuning", 5, tune)
  print("tuning key 5 =", tune)
  for name, cfg in (("vertical   (8192 x 16)", (B * w, h, w, h * w, 1, w)), ("horizontal (2048 x 64)", (B * h, w, 1, w, 0, 1))):
      nseq, T, ic, os_, is_, ts = cfg
      f = lambda: _lib.call("focr_gru_bidir_fwd", K._p(gx), K._p(whh), K._p(bhh), K._p(hseq), K._p(gates), nseq, T, ic, os_, is_, ts, K._stream())
      b = lambda: _lib.call("focr_gru_bidir_bwd", K._p(dh), K._p(whh), K._p(gates), K._p(hseq), K._p(dgx), K._p(dgh), K._p(hprev), nseq, T, ic, os_, is_, ts, K._stream())
      mf, nf = timeit(f); mb, nb = timeit(b)
      print("%s  fwd median %7.1f min %7.1f us (%.2f us/step)   bwd median %7.1f min %7.1f us (%.2f us/step)" % (name, mf, nf, nf / T, mb, nb, nb / T))
print("checksum", float(hseq.double().sum()), float(dgx.double().sum()))

"""per-launch time of the BiLSTM scan kernels (CRNN shape: T=26, B=128, H=256), persistent vs per-step launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib, kernels as K
_lib.load(); _lib.set_precision(2)
t, b, hid = 26, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 256
g = torch.Generator().manual_seed(1)
gx = (torch.rand(t * b, 8 * hid, generator=g) - 0.5).cuda().requires_grad_(True)
whh = ((torch.rand(2, 4 * hid, hid, generator=g) - 0.5) / 8).cuda()
bhh = ((torch.rand(2, 4 * hid, generator=g) - 0.5) / 8).cuda()
gy = (torch.rand(t, b, 2 * hid, generator=g) - 0.5).cuda()
for persistent in (1, 2, 0, 1):       # 1 = default (releases inside the XCD when a group shares one), 2 = always agent scope
    _lib.call("focr_set_tuning", 2, persistent)
    for phase in ("fwd", "fwd+bwd"):
        for _ in range(3):
            y = K.lstm_recurrence(gx, whh, bhh, t, b, b, 1)
            if phase != "fwd": y.backward(gy)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            y = K.lstm_recurrence(gx, whh, bhh, t, b, b, 1)
            if phase != "fwd": y.backward(gy)
        torch.cuda.synchronize()
        print("persistent=%d %-8s %.1f us per call (%.2f us per time step)" % (persistent, phase, (time.perf_counter() - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6 / t))

"""Where the ~8 us per time step of the persistent BiLSTM scan go ("so where?").
Needs the phase-stamp build of the library:
    python __graft_entry__.py --variant lptrace LP_TRACE          (on the authoring box; the .so travels)
    FOCR_LIB=fudanocr_amd/libfocr_hip_lptrace.so python tools/dev/lstm_phases.py [batch]
Thread 0 of every block stamps the 100 MHz wall clock at eight phase boundaries of every forward time step
(csrc/rnn.hip LP_STAMP).  Prints, for the CRNN shape (T = 26, H = 256), per phase: mean / median / p90 over blocks and
steps >= 1, the per-step period, and -- from the partners' stamps -- how long the LAST partner's release precedes the
moment a block sees the counter (the visibility latency of the exchange itself)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from fudanocr_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
if not hasattr(lib, "focr_lstm_trace_dump"):
    raise SystemExit("this library has no phase stamps: build the LP_TRACE variant and set FOCR_LIB (see the docstring)")
_lib.set_precision(2)
t, b, hid = 26, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 256
g = torch.Generator().manual_seed(1)
gx = (torch.rand(t * b, 8 * hid, generator=g) - 0.5).cuda()
whh = ((torch.rand(2, 4 * hid, hid, generator=g) - 0.5) / 8).cuda()
bhh = ((torch.rand(2, 4 * hid, generator=g) - 0.5) / 8).cuda()
NB, NS = 256, 32
for _ in range(5):
    y = K.lstm_recurrence(gx, whh, bhh, t, b, b, 1)
torch.cuda.synchronize()
stamps = np.zeros(NB * NS * 8, dtype=np.uint64)
xcc = np.zeros(NB, dtype=np.uint32)
rc = lib.focr_lstm_trace_dump(stamps.ctypes.data_as(ctypes.c_void_p), xcc.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
ngroups = (b + 31) // 32 * 2
nblk = 8 * ngroups
s = stamps.reshape(NB, NS, 8)[:nblk, :t].astype(np.int64) * 10        # ns (100 MHz wall clock)
names = ["0>1 wait for the partners' counter + L1 invalidate + barrier", "1>2 h(t-1) tile: 4 coalesced 16-byte loads per thread -> LDS + barrier",
         "2>3 16 LDS fragment reads (h) + 24 MFMAs (W fragments in registers)", "3>4 fold the two K halves through LDS (2 barriers)",
         "4>5 gates / c / h + payload staged in LDS (2 barriers)", "5>6 payload stores issued (8 B per thread)",
         "6>7 barrier + counter increment (XCD-local when the census allows, else agent-scope release = L2 write-back)"]
print("persistent LSTM forward, T = %d, B = %d: %d blocks in %d groups of 8; XCC ids per group: %s"
      % (t, b, nblk, ngroups, [sorted(set(int(x) & 15 for x in xcc[gi:nblk:ngroups])) for gi in range(ngroups)]))
d = np.diff(s[:, 1:, :], axis=2)                                       # steps >= 1: [blk, step, 7]
for k, n in enumerate(names):
    v = d[:, :, k].ravel() / 1e3
    print("  %-78s mean %5.2f us  median %5.2f  p90 %5.2f" % (n, v.mean(), np.median(v), np.percentile(v, 90)))
tail = (s[:, 2:, 0] - s[:, 1:-1, 7]).ravel() / 1e3                     # end of step k (stamp 7) -> begin of step k + 1
print("  %-78s mean %5.2f us  median %5.2f  p90 %5.2f" % ("7>0' fp32 outputs (gates, c, h) stored, next step's gx requested",
                                                          tail.mean(), np.median(tail), np.percentile(tail, 90)))
period = np.diff(s[:, 1:, 0], axis=1).ravel() / 1e3
print("  step period %.2f us (median %.2f); whole scan %.1f us" % (period.mean(), np.median(period),
                                                                 (s[:, -1, 7].max() - s[:, 0, 0].min()) / 1e3))
# the exchange itself: for group gi, step k: last partner's stamp 7 of step k - 1 -> this block's stamp 1 of step k
lat, skew = [], []
for gi in range(ngroups):
    blk = s[gi:nblk:ngroups]                                           # the 8 partners (same id mod ngroups)
    for k in range(1, t):
        rel = blk[:, k - 1, 7].max()
        lat.extend((blk[:, k, 1] - rel) / 1e3)
        skew.append((blk[:, k - 1, 7].max() - blk[:, k - 1, 7].min()) / 1e3)
lat, skew = np.array(lat), np.array(skew)
print("  last partner's release -> counter seen + block released: mean %.2f us  median %.2f  p90 %.2f" %
      (lat.mean(), np.median(lat), np.percentile(lat, 90)))
print("  skew between the first and the last of the 8 partners reaching their release: mean %.2f us  p90 %.2f" %
      (skew.mean(), np.percentile(skew, 90)))

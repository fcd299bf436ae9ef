#!/usr/bin/env python3
"""Un-profiled cost of the STN head inside a step (B = 128): forward and backward of STNHead alone, event-timed with the
host running ahead (all launches of 20 iterations enqueued back to back)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fudanocr_amd import kernels as K   # noqa: E402
from fudanocr_amd.model.stn_head import STNHead   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(0)
m = STNHead(3, 20).cuda().train()
x = torch.rand(B, 16, 64, 3, device="cuda")
step = K.StepContext()


def run(n, bwd):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    big = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)
    for _ in range(200):                           # ~60 ms of queued GPU work: the host finishes enqueuing the timed
        big.zero_()                                # region before the GPU reaches it
    a.record()
    for _ in range(n):
        with K.use_context(step):
            feat, pts = m(x)
            if bwd:
                (pts.sum() + feat.sum()).backward()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for _ in range(3):
    run(3, True)
print("STN head B=%d: forward %.1f us, forward + backward %.1f us (event-timed, host ahead)" % (B, run(10, False), run(10, True)))

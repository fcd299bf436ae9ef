#!/usr/bin/env python3
"""How much of the 1e-3 parity budget does the HIP path use?  Compares, on the same inputs and
weights (B=4, dropout off): GPU fp32 (HIP kernels), CPU fp32 oracle, CPU fp64 oracle ("truth")."""
import sys

import torch

sys.path.insert(0, ".")
from fudanocr_amd.smoke import build_models            # noqa: E402
from fudanocr_amd.utils.synth import make_batch        # noqa: E402
from fudanocr_amd.utils.weight_fill import fill_dict_  # noqa: E402
from oracle import sr_oracle as O                      # noqa: E402

lr, hr, _ = make_batch(4, 1234)
pe0 = O.positional_encoding_2d
for arch in ("tbsrn", "tsrn"):
    for training in (False, True):
        net, _, _ = build_models(torch.device("cuda:0"), arch, with_crnn=False)
        net.train(training)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        with torch.no_grad():
            gpu = net(lr.cuda()).cpu().double()
        outs = {}
        for dt in (torch.float32, torch.float64):
            P = O.make_params(O.schema_sr(arch))
            fill_dict_({k: v.data for k, v in P.items()})
            P = {k: (v.detach().to(dt) if v.is_floating_point() else v) for k, v in P.items()}
            O.positional_encoding_2d = lambda *a: pe0(*a).to(dt)
            with torch.no_grad():
                outs[dt] = O.sr_forward(P, arch, lr.to(dt), training).double()
        O.positional_encoding_2d = pe0
        truth = outs[torch.float64]
        mx = truth.abs().max().item()
        print("%-5s train=%-5s  max|sr|=%.3f   GPU-vs-fp64 %.2e   CPU32-vs-fp64 %.2e   GPU-vs-CPU32 %.2e  (rel to max)"
              % (arch, training, mx, (gpu - truth).abs().max().item() / mx,
                 (outs[torch.float32] - truth).abs().max().item() / mx,
                 (gpu - outs[torch.float32]).abs().max().item() / mx))

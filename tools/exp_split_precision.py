#!/usr/bin/env python3
"""Numerics experiment (CPU, oracle): end-to-end SR-pixel error of split-precision contractions.
Every conv / linear / attention matmul of the oracle is replaced by an emulation of
   a*b ~= sum of selected (a_i * b_j) partial products, a = a_0 + a_1 (+ a_2), parts rounded to `dt`,
accumulated in fp32 -- what an MFMA pipeline with `dt` operands and fp32 accumulators would compute.
Compared against the fp64 oracle on the B=4 golden batch in TRAIN mode (the noisiest setting)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from fudanocr_amd.utils.synth import make_batch        # noqa: E402
from fudanocr_amd.utils.weight_fill import fill_dict_  # noqa: E402
from oracle import sr_oracle as O                      # noqa: E402

torch.set_num_threads(8)


def split(x, dt, parts):
    out, r = [], x
    for _ in range(parts):
        p = r.to(dt).float()
        out.append(p)
        r = r - p
    return out


def make_ops(dt, parts, terms):
    """terms: list of (i, j) index pairs of partial products to keep."""
    def bil(fn, a, b):
        aa, bb = split(a, dt, parts), split(b, dt, parts)
        return sum(fn(aa[i], bb[j]) for i, j in terms)

    def conv(P, prefix, x, pad):
        return bil(lambda u, w: F.conv2d(u, w, None, padding=pad), x, P[prefix + "weight"]) + \
            P[prefix + "bias"].view(1, -1, 1, 1)

    def linear(P, prefix, x):
        return bil(lambda u, w: u @ w.t(), x, P[prefix + "weight"]) + P[prefix + "bias"]

    def attention_core(q, k, v, dropout_p=0.0):
        s = bil(lambda u, w: u @ w.transpose(-2, -1), q, k) / (q.shape[-1] ** 0.5)
        p = torch.softmax(s, -1)
        return bil(lambda u, w: u @ w, p, v)
    return conv, linear, attention_core


def run(arch, ops=None):
    saved = (O.conv, O.linear, O.attention_core)
    if ops:
        O.conv, O.linear, O.attention_core = ops
    P = O.make_params(O.schema_sr(arch))
    fill_dict_({k: v.data for k, v in P.items()})
    lr, hr, _ = make_batch(4, 1234)
    with torch.no_grad():
        sr = O.sr_forward(P, arch, lr, True)
    O.conv, O.linear, O.attention_core = saved
    return sr.double()


def truth(arch):
    pe0 = O.positional_encoding_2d
    O.positional_encoding_2d = lambda *a: pe0(*a).double()
    P = O.make_params(O.schema_sr(arch))
    fill_dict_({k: v.data for k, v in P.items()})
    P = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in P.items()}
    lr, _, _ = make_batch(4, 1234)
    with torch.no_grad():
        sr = O.sr_forward(P, arch, lr.double(), True)
    O.positional_encoding_2d = pe0
    return sr


T2 = [(0, 0), (0, 1), (1, 0)]
T2F = T2 + [(1, 1)]
T3 = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
for arch in ("tbsrn", "tsrn"):
    t = truth(arch)
    mx = t.abs().max().item()
    rel = lambda x: (x - t).abs().max().item() / mx
    print("%s  fp32 oracle                         %.2e" % (arch, rel(run(arch))))
    for name, dt, parts, terms in (("bf16 x1 (plain)", torch.bfloat16, 1, [(0, 0)]),
                                   ("bf16 hi/lo, 3 products", torch.bfloat16, 2, T2),
                                   ("bf16 3-way, 6 products", torch.bfloat16, 3, T3),
                                   ("fp16 x1 (plain)", torch.float16, 1, [(0, 0)]),
                                   ("fp16 hi/lo, 3 products", torch.float16, 2, T2),
                                   ("fp16 hi/lo, 4 products", torch.float16, 2, T2F)):
        print("%s  %-34s  %.2e" % (arch, name, rel(run(arch, make_ops(dt, parts, terms)))))

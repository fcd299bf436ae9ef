#!/usr/bin/env python3
"""Per-site precision ablation of the TBSRN FORWARD (CPU, oracle arithmetic emulation).

Baseline = every contraction as bf16 hi/lo split operands, three products, fp32 accumulation ("bf16x3", what the HIP
forward computes).  ONE site at a time then drops products; reported: max |SR - SR_fp64| / max |SR_fp64| on the B = 4
golden batch in TRAIN mode (batch statistics: the noisiest setting; the parity gate is 1e-3, a site is adoptable only
if it stays <= 3e-4).  Sites:
  pv_p_hi     P.V with P rounded to ONE bf16 (P_hi V_hi + P_hi V_lo): no lo plane of P, 2 of 3 MFMAs
  pv_1        P.V as one product (P_hi V_hi)
  qk_2        Q.K^T with 2 products (Q_hi K_hi + Q_lo K_hi)
  qk_1        Q.K^T as one product
  conv3_1     the 3x3 / C = 64 convolutions (halo kernel) as one product
  lin128_1    the K <= 128 transformer linears (Q/K/V/O, FFN, 128 -> 64) as one product
usage: python tools/exp_precision_sites.py [out.md] [--qk-gain G]
  --qk-gain G: scale the q / k projection weights of every block by G first (scores x G^2: sharper attention, the
  sensitivity run quoted in profiles/r03_precision_sites.md)"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from fudanocr_amd.utils.synth import make_batch        # noqa: E402
from fudanocr_amd.utils.weight_fill import fill_dict_  # noqa: E402
from oracle import sr_oracle as O                      # noqa: E402

torch.set_num_threads(8)
BF = torch.bfloat16
QK_GAIN = 1.0


def fill(d):
    fill_dict_(d)
    if QK_GAIN != 1.0:
        for k, v in d.items():
            if ".multihead.linears.0.weight" in k or ".multihead.linears.1.weight" in k:
                v.mul_(QK_GAIN)
T3 = [(0, 0), (0, 1), (1, 0)]


def split(x):
    hi = x.to(BF).float()
    return [hi, (x - hi).to(BF).float()]


def bil(fn, a, b, terms):
    aa, bb = split(a), split(b)
    return sum(fn(aa[i], bb[j]) for i, j in terms)


def make_ops(site):
    def conv(P, prefix, x, pad):
        w = P[prefix + "weight"]
        terms = T3
        if site == "conv3_1" and w.shape[2] == 3 and w.shape[1] % 64 == 0 and w.shape[0] % 64 == 0:
            terms = [(0, 0)]
        return bil(lambda u, v: F.conv2d(u, v, None, padding=pad), x, w, terms) + P[prefix + "bias"].view(1, -1, 1, 1)

    def linear(P, prefix, x):
        w = P[prefix + "weight"]
        terms = [(0, 0)] if (site == "lin128_1" and w.shape[1] <= 128 and "feature_enhancer" in prefix) else T3
        return bil(lambda u, v: u @ v.t(), x, w, terms) + P[prefix + "bias"]

    def attention_core(q, k, v, dropout_p=0.0):
        tq = {"qk_2": [(0, 0), (1, 0)], "qk_1": [(0, 0)]}.get(site, T3)
        s = bil(lambda u, w: u @ w.transpose(-2, -1), q, k, tq) / (q.shape[-1] ** 0.5)
        p = torch.softmax(s, -1)
        tp = {"pv_p_hi": [(0, 0), (0, 1)], "pv_1": [(0, 0)]}.get(site, T3)
        return bil(lambda u, w: u @ w, p, v, tp)
    return conv, linear, attention_core


def run(arch, site):
    saved = (O.conv, O.linear, O.attention_core)
    O.conv, O.linear, O.attention_core = make_ops(site)
    P = O.make_params(O.schema_sr(arch))
    fill({k: v.data for k, v in P.items()})
    lr, hr, _ = make_batch(4, 1234)
    with torch.no_grad():
        sr = O.sr_forward(P, arch, lr, True)
    O.conv, O.linear, O.attention_core = saved
    return sr.double()


def truth(arch):
    pe0 = O.positional_encoding_2d
    O.positional_encoding_2d = lambda *a: pe0(*a).double()
    P = O.make_params(O.schema_sr(arch))
    fill({k: v.data for k, v in P.items()})
    P = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in P.items()}
    lr, _, _ = make_batch(4, 1234)
    with torch.no_grad():
        sr = O.sr_forward(P, arch, lr.double(), True)
    O.positional_encoding_2d = pe0
    return sr


def main():
    global QK_GAIN
    if "--qk-gain" in sys.argv:
        i = sys.argv.index("--qk-gain")
        QK_GAIN = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    t = truth("tbsrn")
    mx = t.abs().max().item()
    rows = []
    for site in ("baseline", "pv_p_hi", "pv_1", "qk_2", "qk_1", "conv3_1", "lin128_1"):
        e = (run("tbsrn", site) - t).abs()
        rows.append((site, e.max().item() / mx, e.mean().item() / mx))
        print("%-10s max %.2e  mean %.2e" % rows[-1], flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("| site (everything else bf16x3) | max err / max | mean err / max | adoptable (<= 3e-4) |\n|---|---|---|---|\n")
            for s, a, b in rows:
                f.write("| %s | %.2e | %.2e | %s |\n" % (s, a, b, "baseline" if s == "baseline" else ("yes" if a <= 3e-4 else "no")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Layer-by-layer deviation of the HIP TBSRN (train mode, dropout off, B=4) from the fp64 oracle."""
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from fudanocr_amd import _lib
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
from fudanocr_amd.utils.weight_fill import fill_dict_
from oracle import sr_oracle as O

if "--fp32" in sys.argv:
    _lib.set_precision(0)
arch = "tbsrn"
lr, hr, _ = make_batch(4, 1234)
net, _, _ = build_models(torch.device("cuda:0"), arch, with_crnn=False)
net.train()
for m in net.modules():
    if isinstance(m, torch.nn.Dropout):
        m.eval()
cap = {}
def hook(name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        cap[name] = o.detach().cpu().double()
    return f
net.stn_head.register_forward_hook(lambda m, i, o: cap.__setitem__("ctrl", o[1].detach().cpu().double()))
net.tps.register_forward_hook(hook("warp"))
net.block1.register_forward_hook(hook("block1"))
for i in range(2, 7):
    getattr(net, "block%d" % i).register_forward_hook(hook("block%d" % i))
    getattr(net, "block%d" % i).bn2.register_forward_hook(hook("block%d.bn2" % i))
    getattr(net, "block%d" % i).bn1.register_forward_hook(hook("block%d.bn1mish" % i))
    fe = getattr(net, "block%d" % i).feature_enhancer
    fe.multihead.register_forward_hook(hook("block%d.mha" % i))
    fe.mul_layernorm1.register_forward_hook(hook("block%d.ln1" % i))
    fe.mul_layernorm3.register_forward_hook(hook("block%d.ln3" % i))
net.block7[1].register_forward_hook(hook("block7+1"))
with torch.no_grad():
    out = net(lr.cuda()).cpu().double()

# ---- fp64 oracle with the same capture points ----
pe0 = O.positional_encoding_2d
O.positional_encoding_2d = lambda *a: pe0(*a).double()
P = O.make_params(O.schema_sr(arch))
fill_dict_({k: v.data for k, v in P.items()})
P = {k: (v.detach().double() if v.is_floating_point() else v) for k, v in P.items()}
ref = {}
with torch.no_grad():
    x = lr.double()
    ctrl = O.stn_head(P, x, True)
    ref["ctrl"] = ctrl
    xw = O.tps_warp(P, x, ctrl)
    ref["warp"] = xw.permute(0, 2, 3, 1)
    b1 = O.conv(P, "block1.0.", xw, 4)
    b1 = torch.where(b1 >= 0, b1, P["block1.1.weight"] * b1)
    ref["block1"] = b1.permute(0, 2, 3, 1)
    h = b1
    for i in range(2, 7):
        p = "block%d." % i
        r = O.conv(P, p + "conv1.", h, 1)
        r = O.mish(O.batchnorm(P, p + "bn1.", r, True))
        ref["block%d.bn1mish" % i] = r.permute(0, 2, 3, 1)
        r = O.batchnorm(P, p + "bn2.", O.conv(P, p + "conv2.", r, 1), True)
        ref["block%d.bn2" % i] = r.permute(0, 2, 3, 1)
        f = p + "feature_enhancer."
        feat = r.reshape(4, 64, -1)
        pe = O.positional_encoding_2d(64, 16, 64).view(1, 64, 1024).expand(4, -1, -1)
        tok = torch.cat([feat, pe], 1).permute(0, 2, 1)
        hd = lambda t: t.view(4, -1, 4, 32).transpose(1, 2)
        q, k, v = (hd(O.linear(P, f + "multihead.linears.%d." % j, tok)) for j in range(3))
        att = O.attention_core(q, k, v).transpose(1, 2).reshape(4, -1, 128)
        att = O.linear(P, f + "multihead.linears.3.", att)
        ref["block%d.mha" % i] = att
        r1 = O.layernorm_std(tok + att, P[f + "mul_layernorm1.a_2"], P[f + "mul_layernorm1.b_2"])
        ref["block%d.ln1" % i] = r1
        ff = O.linear(P, f + "pff.w_2.", F.relu(O.linear(P, f + "pff.w_1.", r1)))
        r3 = O.layernorm_std(r1 + ff, P[f + "mul_layernorm3.a_2"], P[f + "mul_layernorm3.b_2"])
        ref["block%d.ln3" % i] = r3
        h = h + O.linear(P, f + "linear.", r3).permute(0, 2, 1).reshape(h.shape)
        ref["block%d" % i] = h.permute(0, 2, 3, 1)
    h7 = O.batchnorm(P, "block7.1.", O.conv(P, "block7.0.", h, 1), True)
    ref["block7+1"] = (b1 + h7).permute(0, 2, 3, 1)
    full = O.sr_forward(P, arch, lr.double(), True)
print("precision mode", _lib.get_precision())
for k in ref:
    a, b = cap[k].reshape(-1), ref[k].reshape(-1)
    print("%-18s max|ref| %9.3e   max abs err %9.3e   rel-to-max %9.3e" % (k, b.abs().max(), (a - b).abs().max(),
                                                                          (a - b).abs().max() / b.abs().max()))
print("output             rel-to-max %.3e" % ((out - full).abs().max() / full.abs().max()))

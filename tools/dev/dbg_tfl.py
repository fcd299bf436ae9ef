import os, sys, json, types
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from fudanocr_amd import _lib, kernels as K
from fudanocr_amd.loss.text_focus_loss import TextFocusLoss, to_gray_tensor
from fudanocr_amd.loss.transformer import Transformer
from fudanocr_amd.utils.weight_fill import fill_module_
from test_text_focus import make_batch, make_sr, _rel
g = np.load("/root/repo/tests/golden/tfl_step.npz")
for mode in ():
    _lib.set_precision(mode)
    tr = fill_module_(Transformer()).cuda().eval()
    for p in tr.parameters(): p.requires_grad = False
    crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, weight_table=torch.tensor(g["table"]))
    _, hr, labels = make_batch(4, 1234)
    sr = make_sr(hr).cuda().requires_grad_(True)
    loss, mse, att, rec = crit(sr, hr.cuda(), labels)
    d_ce, = torch.autograd.grad(rec, sr, retain_graph=True)
    d = d_ce.cpu()[:, :, ::2, ::4].numpy(); r = g["dsr_ce_sub"]
    print("mode", mode, "rel", _rel(torch.tensor(d), r))
    err = np.abs(d - r)
    print(" per-sample max err", err.reshape(4, -1).max(1), "ref max", np.abs(r).max())
    e2 = err.max(axis=(0, 1))
    print(" rows max", np.round(e2.max(1) / np.abs(r).max(), 3))
    print(" cols max", np.round(e2.max(0) / np.abs(r).max(), 3))
# per-shape dgrad check vs torch on the recognizer's conv shapes
_lib.set_precision(2)
for (h, w, ci, co) in [(32,128,64,128),(16,64,128,256),(8,32,128,256),(8,32,256,256),(8,32,256,512),(8,32,512,512),(8,32,512,1024),(8,32,1024,1024),(16,64,64,128),(32,128,1,64)]:
    x = torch.randn(4, h, w, ci, device="cuda", requires_grad=True)
    wt = (torch.randn(co, ci, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    y = K.conv2d(x, wt, None, pad=(1, 1))
    dy = torch.randn_like(y)
    dx, = torch.autograd.grad(y, x, dy)
    xr = x.detach().permute(0, 3, 1, 2).double().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wt.double(), padding=1)
    dxr, = torch.autograd.grad(yr, xr, dy.permute(0, 3, 1, 2).double())
    print((h, w, ci, co), "fwd", _rel(y.detach().permute(0,3,1,2).cpu(), yr.detach().cpu().float().numpy()), "dgrad", _rel(dx.permute(0,3,1,2).cpu(), dxr.cpu().float().numpy()))

#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by IMPORTING THE REFERENCE (authoring container only).

Recipe = SURVEY.md section 8c: stub the unused `IPython`/`cv2` imports, neutralise the
hard-coded `.cuda()` in tbsrn.py:83, import `model.{tsrn,tbsrn}` and `model.crnn.crnn`
from /root/reference/scene-text-telescope, fill weights with the name-keyed rule
(fudanocr_amd/utils/weight_fill.py), run fp32 CPU forward/backward and store only numeric
inputs-by-seed / expected outputs.  No reference source text is stored.

Run:  python tools/make_golden.py        (needs /root/reference; never runs on the GPU box)
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scene-text-telescope"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from fudanocr_amd.utils.weight_fill import fill_module_   # noqa: E402
from fudanocr_amd.utils.synth import make_batch, ALPHABET  # noqa: E402


def import_reference():
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    sys.modules["IPython"] = ip
    sys.modules["cv2"] = types.ModuleType("cv2")
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from model import tsrn, tbsrn            # noqa
    from model.crnn import crnn              # noqa
    return tsrn, tbsrn, crnn


def schema_of(m):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]


def set_dropout_eval(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.eval()


def grad_norms(m):
    return {k: (float(p.grad.norm()) if p.grad is not None else None) for k, p in m.named_parameters()}


def encode(labels):
    flat = [ALPHABET.index(c) + 1 for s in labels for c in s]
    return torch.tensor(flat, dtype=torch.long), torch.tensor([len(s) for s in labels], dtype=torch.long)


def parse_crnn_data(x):                      # reference interfaces/base.py:319-325 (not importable)
    x = F.interpolate(x, (32, 100), mode="bicubic")
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tsrn, tbsrn, crnn = import_reference()
    os.makedirs(OUT, exist_ok=True)
    B = 4
    lr, hr, labels = make_batch(B, 1234)
    tgt, tlen = encode(labels)

    def build(arch):
        m = tsrn.TSRN(STN=True) if arch == "tsrn" else tbsrn.TBSRN(STN=True)
        fill_module_(m)
        return m

    rec = crnn.CRNN(32, 1, 37, 256)
    fill_module_(rec)
    rec.eval()
    for p in rec.parameters():
        p.requires_grad = False

    schemas = {"tsrn": schema_of(build("tsrn")), "tbsrn": schema_of(build("tbsrn")),
               "crnn": schema_of(rec)}
    with open(os.path.join(OUT, "schema.json"), "w") as f:
        json.dump(schemas, f)

    for arch in ("tsrn", "tbsrn"):
        # ---- F1/F2: train-mode forward + MSE backward (dropout modules in eval) ----
        m = build(arch)
        m.train()
        set_dropout_eval(m)
        x = lr.clone().requires_grad_(True)
        sr = m(x)
        mse = F.mse_loss(sr, hr)
        (mse * 100).backward()
        gn = grad_norms(m)
        sd = m.state_dict()
        np.savez_compressed(
            os.path.join(OUT, "%s_train_mse.npz" % arch),
            sr=sr.detach().numpy(), mse=np.float32(mse.item()),
            dlr=x.grad.numpy(),
            g_block1_w=dict(m.named_parameters())["block1.0.weight"].grad.numpy(),
            g_block8_b=dict(m.named_parameters())["block8.1.bias"].grad.numpy(),
            g_fc2_w=dict(m.named_parameters())["stn_head.stn_fc2.weight"].grad.numpy(),
            g_b2c1_w=dict(m.named_parameters())["block2.conv1.weight"].grad.numpy(),
            bn_rm=sd["block2.bn1.running_mean"].numpy(), bn_rv=sd["block2.bn1.running_var"].numpy(),
        )
        with open(os.path.join(OUT, "%s_train_mse_gradnorms.json" % arch), "w") as f:
            json.dump(gn, f)

        # ---- F3: eval mode (STN skipped, BN running stats) ----
        m = build(arch)
        m.eval()
        with torch.no_grad():
            sr_e = m(lr)
        np.savez_compressed(os.path.join(OUT, "%s_eval.npz" % arch), sr=sr_e.numpy())

        # ---- F5: end-to-end composition SR -> parse_crnn_data -> CRNN -> CTC (+MSE) ----
        m = build(arch)
        m.train()
        set_dropout_eval(m)
        x = lr.clone().requires_grad_(True)
        sr = m(x)
        mse = F.mse_loss(sr, hr)
        logits = rec(parse_crnn_data(sr[:, :3]))
        lp = F.log_softmax(logits, 2)
        ctc = F.ctc_loss(lp, tgt, torch.full((B,), 26, dtype=torch.long), tlen, blank=0,
                         reduction="mean", zero_infinity=True)
        ((mse + ctc) * 100).backward()
        np.savez_compressed(
            os.path.join(OUT, "%s_e2e_ctc.npz" % arch),
            sr=sr.detach().numpy(), mse=np.float32(mse.item()), ctc=np.float32(ctc.item()),
            logits=logits.detach().numpy(), dlr=x.grad.numpy(),
            g_block1_w=dict(m.named_parameters())["block1.0.weight"].grad.numpy(),
        )
        with open(os.path.join(OUT, "%s_e2e_ctc_gradnorms.json" % arch), "w") as f:
            json.dump(grad_norms(m), f)

        # ---- F8: 3-step trajectory (clip 0.25 + Adam(1e-4,(0.5,0.999))), MSE+CTC loss ----
        m = build(arch)
        m.train()
        set_dropout_eval(m)
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, betas=(0.5, 0.999))
        traj = {"loss": [], "mse": [], "ctc": [], "grad_norm": []}
        for step in range(3):
            lr_s, hr_s, lab_s = make_batch(B, 1234 + step)
            t_s, l_s = encode(lab_s)
            sr = m(lr_s)
            mse = F.mse_loss(sr, hr_s)
            lp = F.log_softmax(rec(parse_crnn_data(sr[:, :3])), 2)
            ctc = F.ctc_loss(lp, t_s, torch.full((B,), 26, dtype=torch.long), l_s, blank=0,
                             reduction="mean", zero_infinity=True)
            loss = mse + ctc
            opt.zero_grad()
            (loss * 100).backward()
            gn = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
            opt.step()
            traj["loss"].append(float(loss)); traj["mse"].append(float(mse))
            traj["ctc"].append(float(ctc)); traj["grad_norm"].append(float(gn))
        sd = m.state_dict()
        traj["param_abs_sum"] = {k: float(v.double().abs().sum()) for k, v in sd.items()
                                 if v.is_floating_point()}
        with open(os.path.join(OUT, "%s_traj3.json" % arch), "w") as f:
            json.dump(traj, f)

    # ---- F4: CRNN leg alone ----
    g = torch.Generator().manual_seed(77)
    img = torch.rand(B, 3, 32, 128, generator=g)
    with torch.no_grad():
        gray = parse_crnn_data(img)
        logits = rec(gray)
    np.savez_compressed(os.path.join(OUT, "crnn_leg.npz"), gray=gray.numpy(), logits=logits.numpy())

    # ---- F6: unit vectors for the semantic traps ----
    g = torch.Generator().manual_seed(5)
    ln = tbsrn.LayerNorm(128)
    fill_ln = {"a_2": torch.rand(128, generator=g) + 0.5, "b_2": torch.rand(128, generator=g) - 0.5}
    ln.load_state_dict(fill_ln)
    xln = torch.randn(6, 128, generator=g) * 3 + 1
    mish_in = torch.tensor([-25.0, -5.0, -1.0, -1e-3, 0.0, 1e-3, 1.0, 5.0, 19.9, 20.1, 25.0])
    pe = tbsrn.positionalencoding2d(64, 16, 64)
    tps = build("tsrn").tps
    ctrl = tps.target_control_points[None].repeat(2, 1, 1).clone()
    ctrl[1] += (torch.rand(20, 2, generator=g) - 0.5) * 0.2      # sample 1: perturbed points
    img2 = torch.rand(2, 3, 16, 64, generator=g)
    warped, src = tps(img2, ctrl)
    np.savez_compressed(
        os.path.join(OUT, "units.npz"),
        ln_a=fill_ln["a_2"].numpy(), ln_b=fill_ln["b_2"].numpy(), ln_x=xln.numpy(),
        ln_y=ln(xln).detach().numpy(),
        mish_x=mish_in.numpy(), mish_y=tbsrn.mish()(mish_in).numpy(),
        pe=pe.numpy(),
        tps_ctrl=ctrl.numpy(), tps_img=img2.numpy(), tps_out=warped.numpy(),
        tps_inv=tps.inverse_kernel.numpy(), tps_repr=tps.target_coordinate_repr.numpy(),
    )
    # ---- F9: evaluation metrics (utils/ssim_psnr.py) and label filter (utils/util.py) ----
    from utils import ssim_psnr as ref_metrics
    from utils import util as ref_util
    g = torch.Generator().manual_seed(11)
    ia = torch.rand(3, 3, 32, 128, generator=g)
    ib = (ia + 0.1 * torch.randn(3, 3, 32, 128, generator=g)).clamp(0, 1)
    with open(os.path.join(OUT, "metrics.json"), "w") as f:
        json.dump({"psnr": float(ref_metrics.calculate_psnr(ia, ib)), "ssim": float(ref_metrics.SSIM()(ia, ib)),
                   "str_filt": [[t, v, ref_util.str_filt(t, v)] for t, v in
                                (("Ab-9 z!", "lower"), ("Ab-9 z!", "all"), ("Ab-9 z!", "digit"), ("Ab-9 z!", "upper"))]}, f)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden written to", OUT, "total bytes", total)


if __name__ == "__main__":
    main()

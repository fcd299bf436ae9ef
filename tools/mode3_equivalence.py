#!/usr/bin/env python3
"""Does the bench's default arithmetic (precision mode 3: single-bf16 data-gradient products) TRAIN like the
fp32-equivalent arithmetic (mode 1: split products at every site)?

N optimisation steps of the c3 step (TBSRN + frozen CRNN-CTC, reference step interfaces/super_resolution.py:79-84) on a fixed
cycle of 8 batches at per-GPU batch B, dropout off, from identical weights, three times: mode 1, mode 1 AGAIN (the run-to-run
spread of the SAME arithmetic -- fp32 atomics in a few backward kernels + Adam's normalised steps make two identical runs
drift apart; that spread is the yardstick every other difference is read against) and mode 3.  Reports, per run pair, the
largest relative loss difference over all steps, the worst displacement cosine over the smooth parameter tensors
(final - initial), and the PSNR of the eval-mode forward on a held-out batch.

usage: python tools/mode3_equivalence.py [--steps 300] [--batch 128] [--out profiles/r06_mode3_equivalence.json]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SMOOTH = ("conv1.weight", "conv2.weight", ".pff.w_1.weight", ".pff.w_2.weight", ".linears.0.weight", ".linears.1.weight",
          ".linears.2.weight", ".linears.3.weight", ".feature_enhancer.linear.weight")


def run(mode, steps, batch, cycle=8):
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.smoke import build_models
    from fudanocr_amd.utils.synth import make_batch
    from fudanocr_amd.utils.ssim_psnr import calculate_psnr
    _lib.load()
    old = _lib.get_precision()
    _lib.set_precision(mode)
    try:
        dev = torch.device("cuda", 0)
        net, rec, crit = build_models(dev, "tbsrn", with_crnn=True)
        p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
        step = TrainStep(net, crit, dropout=False, seed=11)
        data = []
        for i in range(cycle):
            lr, hr, labels = make_batch(batch, 4000 + i)
            data.append((lr.to(dev), hr.to(dev), crit.encode(labels, dev)))
        losses = torch.empty(steps, device=dev)
        for s in range(steps):
            lr, hr, enc = data[s % cycle]
            losses[s] = step(lr, hr, encoded=enc)["loss"]
        torch.cuda.synchronize()
        disp = {k: (v.detach() - p0[k]).flatten().double().cpu() for k, v in net.named_parameters()
                if k.startswith("block") and ".gru" not in k and any(k.endswith(s_) for s_ in SMOOTH)}   # (.gru: dead in TBSRN)
        lr, hr, _ = make_batch(batch, 9999)
        net.eval()
        with torch.no_grad():
            sr = net(lr.to(dev))
        psnr = float(calculate_psnr(sr[:, :3], hr.to(dev)[:, :3]))
        return {"losses": losses.double().cpu(), "disp": disp, "psnr": psnr, "replayed": step.recorded is not None}
    finally:
        _lib.set_precision(old)


def compare(a, b):
    rel = ((a["losses"] - b["losses"]).abs() / a["losses"].abs())
    cos = {k: float(torch.dot(a["disp"][k], b["disp"][k]) / (a["disp"][k].norm() * b["disp"][k].norm() + 1e-30))
           for k in a["disp"]}
    worst = min(cos, key=cos.get)
    return {"max_rel_loss_diff": float(rel.max()), "step_of_max": int(rel.argmax()),
            "rel_loss_diff_at": {str(s): float(rel[s]) for s in (0, 9, 49, 99, 199, len(rel) - 1) if s < len(rel)},
            "worst_displacement_cosine": cos[worst], "worst_tensor": worst,
            "mean_displacement_cosine": sum(cos.values()) / len(cos), "psnr_diff_db": abs(a["psnr"] - b["psnr"])}


def main(steps=300, batch=128, out=None, repeats=3):
    runs = {1: [run(1, steps, batch) for _ in range(repeats)], 3: [run(3, steps, batch) for _ in range(repeats)]}

    def pairs(xs, ys, same):
        return [compare(x, y) for i, x in enumerate(xs) for j, y in enumerate(ys) if (not same or i < j)]

    def summary(ps):
        return {"pairs": len(ps),
                "max_rel_loss_diff": max(p["max_rel_loss_diff"] for p in ps),
                "worst_displacement_cosine": min(p["worst_displacement_cosine"] for p in ps),
                "mean_displacement_cosine": sum(p["mean_displacement_cosine"] for p in ps) / len(ps),
                "psnr_diff_db_max": max(p["psnr_diff_db"] for p in ps),
                "psnr_diff_db_mean": sum(p["psnr_diff_db"] for p in ps) / len(ps)}

    a, c = runs[1][0], runs[3][0]
    res = {"workload": "c3 step (TBSRN + frozen CRNN-CTC), B = %d, %d steps on a fixed 8-batch cycle, dropout off, identical "
                       "initial weights, Adam lr 1e-4 + clip 0.25; %d runs per precision mode" % (batch, steps, repeats),
           "replayed": all(r["replayed"] for rs in runs.values() for r in rs),
           "loss_first_last": {"mode1": [float(a["losses"][0]), float(a["losses"][-1])],
                               "mode3": [float(c["losses"][0]), float(c["losses"][-1])]},
           "psnr_db": {"mode1": [r["psnr"] for r in runs[1]], "mode3": [r["psnr"] for r in runs[3]]},
           "within_mode1": summary(pairs(runs[1], runs[1], True)),
           "within_mode3": summary(pairs(runs[3], runs[3], True)),
           "mode1_vs_mode3": summary(pairs(runs[1], runs[3], False)),
           "first_pair_detail": {"mode1_vs_mode1": compare(runs[1][0], runs[1][1]), "mode1_vs_mode3": compare(a, c)}}
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--out", default=None)
    ap.add_argument("--repeats", type=int, default=3)
    a_ = ap.parse_args()
    print(json.dumps(main(a_.steps, a_.batch, a_.out, a_.repeats), indent=1))

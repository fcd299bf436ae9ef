#!/usr/bin/env python3
"""Fixture for the text-gestalt half of row N1 (stroke-focus loss): tests/golden/sfl_*.npz|json, generated
by RUNNING THE REFERENCE (authoring container only):
  * text-gestalt/loss/transformer_english_decomposition.py `Transformer` is imported as a module (torch + numpy only;
    `.cuda()` made an identity) -- the frozen stroke-level recognizer, `.eval()` as build_up_transformer leaves it;
  * loss/stroke_focus_loss.py cannot be imported (cv2; its constructor opens ./dataset/mydata/english_decomposition.txt
    and a checkpoint, both absent), so `to_gray_tensor`, `StrokeFocusLoss.label_stroke_encoder` and
    `StrokeFocusLoss.forward` are compiled from their own AST nodes and executed: the reference's code runs, nothing of
    it is stored.  The absent decomposition file is replaced by the product's seeded stand-in table
    (`fudanocr_amd.loss.stroke_focus_loss.standin_decomposition`) -- an INPUT of the fixture, stored with it.
Weights: name-keyed fill.  Stored: the stroke encoding of the labels, predictions, correct lists, attention maps
(subsampled), the loss values for stroke_lambda = 50 and d loss / d SR image (+ a smooth probe of the map gradient)."""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/text-gestalt"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from fudanocr_amd.utils.weight_fill import fill_module_   # noqa: E402
from fudanocr_amd.utils.synth import make_batch            # noqa: E402
from tools.make_golden_tfl import _fn, make_sr, map_probe  # noqa: E402


def standin_decomposition(seed=2021):
    """the same rule as fudanocr_amd.loss.stroke_focus_loss.standin_decomposition (kept independent of the product)"""
    chars = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    rs = np.random.RandomState(seed)
    return {c: "".join(str(int(d)) for d in rs.randint(1, 10, size=int(rs.randint(1, 5)))) for c in chars}


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)
    from loss import transformer_english_decomposition as RT      # the reference stroke-level recognizer
    sfl_path = os.path.join(REF, "loss", "stroke_focus_loss.py")
    ns = {"torch": torch, "np": np, "time": __import__("time")}
    exec(compile(ast.Module(body=[_fn(sfl_path, "to_gray_tensor")], type_ignores=[]), "<reference helper>", "exec"), ns)
    cls = ast.ClassDef(name="SFL", bases=[], keywords=[], decorator_list=[],
                       body=[_fn(sfl_path, "label_stroke_encoder", "StrokeFocusLoss"),
                             _fn(sfl_path, "forward", "StrokeFocusLoss")])
    ast.fix_missing_locations(cls)
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "<reference StrokeFocusLoss methods>", "exec"), ns)
    sfl = ns["SFL"]()
    sfl.args = types.SimpleNamespace(text_focus=True, stroke_lambda=50.0)
    sfl.mse_loss, sfl.l1_loss = torch.nn.MSELoss(), torch.nn.L1Loss()
    sfl.english_stroke_alphabet = "0123456789"
    sfl.english_stroke_dict = {c: i for i, c in enumerate(sfl.english_stroke_alphabet)}
    sfl.dic = standin_decomposition()
    model = RT.Transformer()
    fill_module_(model)
    model.eval()
    sfl.transformer = model
    schema = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]

    B = 4
    _, hr, labels = make_batch(B, 1234)
    labels = list(labels)
    labels[1] = labels[1][:2] + "#" + labels[1][2:]          # a character without a decomposition is skipped (:55-56)
    sr = make_sr(hr).requires_grad_(True)
    loss, mse, att, rec = sfl.forward(sr, hr, labels)
    assert rec == -1
    length, text_input, text_gt = sfl.label_stroke_encoder(labels)
    datt, = torch.autograd.grad(att, sr, retain_graph=True)
    loss.backward()
    sr2 = sr.detach().clone().requires_grad_(True)
    pred2, amap2, correct_sr = model(ns["to_gray_tensor"](sr2), length, text_input, test=False)
    probe = map_probe(amap2.shape)
    dsr_map, = torch.autograd.grad((amap2 * probe).sum(), sr2)
    with torch.no_grad():
        pred, amap, _ = model(ns["to_gray_tensor"](sr), length, text_input, test=False)
        _, _, correct_hr = model(ns["to_gray_tensor"](hr), length, text_input, test=False)
        # the masked 4-channel input path of the recognizer (:363-367): luma of the first three channels
        rgbm = torch.cat([sr.detach(), torch.ones_like(sr[:, :1])], 1)
        pred4, _, _ = model(rgbm, length, text_input, test=False)
        padded = model(ns["to_gray_tensor"](sr), length, text_input, test=True)
    # a correct_list with both outcomes: feed the recognizer its own greedy continuation for sample 0
    ti2 = text_input.clone()
    n0 = int(length[0])
    with torch.no_grad():
        for j in range(1, n0):       # causal decoder: position j - 1's output only depends on inputs 0 .. j - 1
            ti2[0, j] = model(ns["to_gray_tensor"](sr), length, ti2, test=True)[0, j - 1].argmax(-1)
    with torch.no_grad():
        _, _, correct_mixed = model(ns["to_gray_tensor"](sr), length, ti2, test=False)
    np.savez_compressed(
        os.path.join(OUT, "sfl_step.npz"), pred=pred.numpy(), pred4=pred4.numpy(),
        map_sub=amap[:, ::4, :, ::8].numpy(), dsr_sub=sr.grad[:, :, ::2, ::4].numpy(),
        datt_sub=datt[:, :, ::2, ::4].numpy(), dsr_map_sub=dsr_map[:, :, ::2, ::4].numpy(),
        losses=np.array([loss.item(), mse.item(), att.item(), -1.0], dtype=np.float64))
    with open(os.path.join(OUT, "sfl_schema.json"), "w") as f:
        json.dump({"schema": schema, "labels": labels, "decomposition": sfl.dic, "length": length.tolist(),
                   "text_input": text_input.tolist(), "text_gt": text_gt.tolist(),
                   "correct_hr": [bool(v) for v in correct_hr], "correct_sr": [bool(v) for v in correct_sr],
                   "text_input_mixed": ti2.tolist(), "correct_mixed": [bool(v) for v in correct_mixed],
                   "stroke_lambda": 50.0}, f)
    print("sfl fixture: losses", loss.item(), mse.item(), att.item(), "lengths", length.tolist(), "correct hr/sr/mixed",
          [bool(v) for v in correct_hr], [bool(v) for v in correct_sr], [bool(v) for v in correct_mixed],
          "params", sum(p.numel() for p in model.parameters()))


if __name__ == "__main__":
    main()

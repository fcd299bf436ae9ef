"""Row N2 (BASELINE configs[4]): stroke-level-decomposition transformer recognizer.
CPU: the oracle restatement against fixture F7 (generated from the imported reference, tools/make_golden_sld.py).
GPU (-m gpu): the HIP product model against the fixture and the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from fudanocr_amd.sld.synth import make_sld_batch
from fudanocr_amd.utils.weight_fill import fill_dict_


def _rel(got, ref):
    got, ref = torch.as_tensor(np.asarray(got)).double(), torch.as_tensor(np.asarray(ref)).double()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def _fx(golden_dir):
    return (np.load(os.path.join(golden_dir, "sld_step.npz")), json.load(open(os.path.join(golden_dir, "sld_schema.json"))),
            json.load(open(os.path.join(golden_dir, "sld_traj2.json"))),
            json.load(open(os.path.join(golden_dir, "sld_gradnorms.json"))))


def test_sld_oracle_matches_reference_fixture(golden_dir):
    from oracle import sld_oracle as O
    g, sc, traj, gn = _fx(golden_dir)
    P = O.make_params()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in P.items()] == sc["schema"]
    fill_dict_({k: v.data for k, v in P.items()})
    image, _ = make_sld_batch(4, 1234)
    length, text_input, text_gt = O.converter_stroke(sc["labels"], sc["table"])
    assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
    opt = O.AdadeltaState([v for k, v in P.items() if v.requires_grad])
    out = O.train_step(P, opt, image, length, text_input, text_gt)
    assert _rel(out["pred"], g["pred"]) < 1e-4
    assert abs(out["loss"] - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert _rel(out["conv"][:, ::64, ::4, ::4], g["conv_sub"]) < 1e-4
    assert _rel(out["map"][:, :, :, ::16], g["map_sub"]) < 1e-4
    assert _rel(P["encoder.bn1.running_mean"], g["bn1_rm"]) < 1e-5 and _rel(P["encoder.bn1.running_var"], g["bn1_rv"]) < 1e-5
    assert abs(out["grad_norm"] - traj["grad_norm"][0]) < 1e-3 * traj["grad_norm"][0]
    out2 = O.train_step(P, opt, image, length, text_input, text_gt)
    assert abs(out2["loss"] - traj["loss"][1]) < 2e-2 * traj["loss"][1]       # Adadelta(lr=1) steps ~ sign(g)*3e-3
    dead = [k for k, v in gn.items() if v is None]
    assert sorted(dead) == sorted(k for k in P if "compress_attention_linear" in k)

#!/usr/bin/env python3
"""Fixture for row a21 (greedy CTC decode): tests/golden/decode.json, produced by RUNNING THE REFERENCE's own decoders
(authoring container only, /root/reference never travels):
  * utils/utils_crnn.py strLabelConverter.decode (imported as a module), raw and collapsed, single and batched;
  * interfaces/super_resolution.py TextSR.get_crnn_pred -- `interfaces` cannot be imported here (torchvision, lmdb,
    easydict, tensorboard are absent), so the method's own AST node is compiled and called: the reference's code runs,
    nothing of it is stored.
Inputs: the CRNN logits of fixture F4 (crnn_leg.npz) plus seeded index sequences with runs, blanks and edge cases.
Only inputs-by-value and the decoded strings are written."""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scene-text-telescope"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def main():
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    sys.modules["IPython"] = ip
    sys.path.insert(0, REF)
    from utils import utils_crnn as ref_codec                       # noqa: E402
    src = open(os.path.join(REF, "interfaces", "super_resolution.py")).read()
    fn = next(n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "get_crnn_pred")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference get_crnn_pred>", "exec"), ns)
    get_crnn_pred = ns["get_crnn_pred"]
    conv = ref_codec.strLabelConverter("0123456789abcdefghijklmnopqrstuvwxyz")

    g = torch.Generator().manual_seed(2024)
    seqs = []
    for _ in range(12):                                             # runs of repeated symbols and blanks
        t, row = 26, []
        while len(row) < t:
            sym = int(torch.randint(0, 37, (1,), generator=g)) if torch.rand(1, generator=g) > 0.35 else 0
            row += [sym] * int(torch.randint(1, 4, (1,), generator=g))
        seqs.append(row[:t])
    seqs.append([0] * 26)                                           # all blank -> ''
    seqs.append([5] * 26)                                           # one long run -> one character
    seqs.append([1, 0, 1, 0, 1] + [0] * 21)                         # blank-separated repeats survive
    seqs.append(list(range(1, 27)))                                 # no repeats, no blanks
    idx = torch.tensor(seqs, dtype=torch.long)                      # [N, T]
    scores = torch.nn.functional.one_hot(idx, 37).float() * 5 + torch.rand(idx.shape + (37,), generator=g)
    logits = torch.tensor(np.load(os.path.join(OUT, "crnn_leg.npz"))["logits"])      # [26, 4, 37]
    out = {"index_rows": seqs, "cases": []}
    for name, sc in (("synthetic", scores), ("crnn_leg", logits.permute(1, 0, 2).contiguous())):
        preds = sc.argmax(2)                                        # [N, T]
        n, t = preds.shape
        flat = preds.reshape(-1).to(torch.int32)
        sizes = torch.IntTensor([t] * n)
        out["cases"].append({
            "name": name,
            "argmax": preds.tolist(),
            "decode": conv.decode(flat, sizes, raw=False),
            "decode_raw": conv.decode(flat, sizes, raw=True),
            "decode_single": [conv.decode(preds[i].to(torch.int32), torch.IntTensor([t]), raw=False) for i in range(n)],
            "get_crnn_pred": get_crnn_pred(None, sc),
            # top-2 margin per frame: consumers comparing strings from recomputed logits skip near-ties
            "min_margin": float((sc.topk(2, dim=2).values[..., 0] - sc.topk(2, dim=2).values[..., 1]).min()),
        })
    with open(os.path.join(OUT, "decode.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(OUT, "decode_scores.npz"), synthetic=scores.numpy())
    print("wrote decode.json:", [c["decode"][:4] for c in out["cases"]])


if __name__ == "__main__":
    main()

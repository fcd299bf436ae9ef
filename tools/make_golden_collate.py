#!/usr/bin/env python3
"""Fixture for row N3 (input pipeline): tests/golden/collate.npz, produced by RUNNING THE REFERENCE's
dataset/dataset.py (`alignCollate_real`, `alignCollate_syn`, `resizeNormalize` incl. the mean-threshold mask) on
seeded synthetic PIL images (authoring container only).  The reference module imports `lmdb` and
`torchvision.transforms`, neither of which is in this image: `lmdb` is stubbed (unused by the collate classes) and
`ToTensor` is stubbed with its documented semantics for 8-bit PIL images (HWC uint8 -> CHW float32 / 255).  PIL -- the
third-party code that does the actual resampling -- is the same library on both sides.
Stored: the input pixel arrays and the collated batches as uint8 (the outputs are exactly k/255)."""
import os
import sys
import types

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/scene-text-telescope"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True


def synth_images(seed, sizes):
    """smooth colour fields with a few dark 'strokes': compressible and bicubic-relevant"""
    rng = np.random.RandomState(seed)
    out = []
    for (w, h) in sizes:
        low = rng.randint(40, 255, (max(2, h // 8), max(2, w // 8), 3)).astype(np.uint8)
        img = np.asarray(Image.fromarray(low).resize((w, h), Image.BILINEAR)).copy()
        for _ in range(6):
            x0, y0 = rng.randint(0, w - 4), rng.randint(0, h - 4)
            img[y0:y0 + rng.randint(2, h // 2), x0:x0 + rng.randint(1, 5)] = rng.randint(0, 60, 3)
        out.append(img)
    return out


def main():
    ip = types.ModuleType("IPython")
    ip.embed = lambda *a, **k: None
    sys.modules["IPython"] = ip
    sys.modules["lmdb"] = types.ModuleType("lmdb")
    tv, tr = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms")

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img, dtype=np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)
    tr.ToTensor = ToTensor
    tv.transforms = tr
    sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
    sys.path.insert(0, REF)
    from dataset import dataset as D                                  # the reference module

    hr_sizes = [(140, 40), (128, 32), (301, 77), (97, 33), (256, 64)]
    lr_sizes = [(70, 20), (64, 16), (150, 38), (49, 17), (128, 32)]
    hr_np, lr_np = synth_images(1, hr_sizes), synth_images(2, lr_sizes)
    words = ["Hello-World", "abc123", "FUDAN ocr!", "x", "TextZoom_2020"]
    batch = [(Image.fromarray(a), Image.fromarray(b), w) for a, b, w in zip(hr_np, lr_np, words)]
    out = {}
    for mask in (False, True):
        hr, lr, labels = D.alignCollate_real(imgH=32, imgW=128, down_sample_scale=2, mask=mask)(batch)
        assert labels == tuple(words)
        for name, t in (("hr", hr), ("lr", lr)):
            u8 = (t * 255).round().to(torch.uint8)
            assert torch.equal(u8.float().div(255), t)              # outputs are exactly k / 255
            out["real_%s_mask%d" % (name, int(mask))] = u8.numpy()
    hr, lr, _ = D.alignCollate_syn(imgH=32, imgW=128, down_sample_scale=2, mask=True)([(b[0], b[2]) for b in batch])
    out["syn_hr"], out["syn_lr"] = (hr * 255).round().to(torch.uint8).numpy(), (lr * 255).round().to(torch.uint8).numpy()
    for i, (a, b) in enumerate(zip(hr_np, lr_np)):
        out["in_hr_%d" % i], out["in_lr_%d" % i] = a, b
    np.savez_compressed(os.path.join(OUT, "collate.npz"), **out)
    import json
    with open(os.path.join(OUT, "collate_labels.json"), "w") as f:
        json.dump({"words": words, "filtered": {v: [D.str_filt(w, v) for w in words] for v in ("all", "lower", "upper", "digit")}}, f)
    print("wrote collate.npz", os.path.getsize(os.path.join(OUT, "collate.npz")), "bytes")


if __name__ == "__main__":
    main()

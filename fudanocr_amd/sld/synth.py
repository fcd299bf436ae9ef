"""Seeded synthetic batches of the stroke-level-decomposition recognizer:
image ~ U[-1,1) [B,3,32,32] (the reference's resizeNormalize maps pixels to [-1,1], SLD data/lmdbReader.py:85);
labels: stroke strings over '12345' of length 1..29 (train.py feeds character_to_strokelist[ch] + '$'; here the
stroke string itself is drawn, which is what the converter sees after that lookup)."""
import torch

ALPHABET_STROKE = "<12345$"


def make_sld_batch(batch, seed=1234, size=32):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    image = torch.randint(0, 1 << 24, (batch, 3, size, size), generator=g).to(torch.float32) / float(1 << 23) - 1.0
    lens = torch.randint(1, 30, (batch,), generator=g)
    labels = []
    for n in lens.tolist():
        idx = torch.randint(1, 6, (n,), generator=g).tolist()
        labels.append("".join(ALPHABET_STROKE[i] for i in idx))
    return image, labels

"""Seeded synthetic TextZoom-shaped batches (SURVEY.md section 8d / BASELINE.md section 3).

lr ~ U[0,1) [B,C,16,64], hr ~ U[0,1) [B,C,32,128] (ToTensor range, reference
dataset/dataset.py:140,145); labels: length ~ U{3..10}, characters uniform over '0-9a-z'.
Generated on the CPU generator so every consumer (golden script, oracle, HIP tests,
bench) sees identical data; callers move the tensors to the device.
"""
import torch

ALPHABET = "0123456789abcdefghijklmnopqrstuvwxyz"


def make_batch(batch, seed=1234, in_planes=3, height=16, width=64, scale=2):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    # integer draws / 2^24: exact in fp32 and bit-identical on every host (see weight_fill._u)
    lr = torch.randint(0, 1 << 24, (batch, in_planes, height, width), generator=g).to(torch.float32) / float(1 << 24)
    hr = torch.randint(0, 1 << 24, (batch, in_planes, height * scale, width * scale),
                       generator=g).to(torch.float32) / float(1 << 24)
    lens = torch.randint(3, 11, (batch,), generator=g)
    labels = []
    for n in lens.tolist():
        idx = torch.randint(0, len(ALPHABET), (n,), generator=g).tolist()
        labels.append("".join(ALPHABET[i] for i in idx))
    return lr, hr, labels

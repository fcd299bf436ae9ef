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


def with_mask(img):
    """append the `--mask` channel to an RGB batch the way the reference's resizeNormalize(mask=True) derives it from
    the image (dataset/dataset.py:146-151): luma, threshold at the image's mean luma, 1.0 where NOT brighter.
    (Exact-in-fp32 integer luma, so every host builds the identical 4-channel synthetic batch.)"""
    u8 = torch.floor(img * 255 + 0.5)
    luma = torch.floor((u8[:, 0] * 299 + u8[:, 1] * 587 + u8[:, 2] * 114 + 500) / 1000)
    thr = luma.mean(dim=(1, 2), keepdim=True)
    return torch.cat([img, (luma <= thr).to(img.dtype).unsqueeze(1)], 1)

"""PSNR / SSIM evaluation metrics with the reference's definitions
(scene-text-telescope/utils/ssim_psnr.py:9-15 PSNR on the first 3 channels in [0,1]*255;
:18-78 SSIM: 11x11 Gaussian window sigma 1.5, C1=0.01^2, C2=0.03^2, per-channel depthwise).
Evaluation-only host code: torch device ops, not part of the timed step."""
import math

import torch
import torch.nn.functional as F


def calculate_psnr(img1, img2):
    mse = ((img1[:, :3] * 255 - img2[:, :3] * 255) ** 2).mean()
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


def _window(size, sigma, channel, device, dtype):
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2.0 * sigma ** 2)) for x in range(size)], dtype=dtype)
    g = (g / g.sum()).unsqueeze(1)
    return (g @ g.t()).expand(channel, 1, size, size).contiguous().to(device)


class SSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average = window_size, size_average

    def forward(self, img1, img2):
        img1, img2 = img1[:, :3], img2[:, :3]
        c = img1.shape[1]
        w = _window(self.window_size, 1.5, c, img1.device, img1.dtype)
        pad = self.window_size // 2
        blur = lambda t: F.conv2d(t, w, padding=pad, groups=c)
        mu1, mu2 = blur(img1), blur(img2)
        s11 = blur(img1 * img1) - mu1 * mu1
        s22 = blur(img2 * img2) - mu2 * mu2
        s12 = blur(img1 * img2) - mu1 * mu2
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
        return m.mean() if self.size_average else m.mean(1).mean(1).mean(1)

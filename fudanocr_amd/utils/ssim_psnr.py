"""PSNR / SSIM evaluation metrics with the reference's definitions
(scene-text-telescope/utils/ssim_psnr.py:9-15 PSNR on the first 3 channels in [0,1]*255;
:18-78 SSIM: 11x11 Gaussian window sigma 1.5, C1=0.01^2, C2=0.03^2, per-channel depthwise).

CUDA tensors (the harness' eval path, interfaces/super_resolution.py:178-181) go through ONE fused HIP pass
(csrc/eval_metrics.hip: squared-error sum and SSIM-map sum per image, fixed-order reduction) -- there is no torch
fallback for them.  CPU tensors (host-side evaluation of saved images, the CPU golden test) use the same formulas
written with torch ops."""
import ctypes
import math

import torch
import torch.nn.functional as F


def _gaussian(size, sigma=1.5):
    """the reference's 1-D window, built the same way (python floats -> fp32 tensor -> fp32 normalisation)"""
    g = torch.Tensor([math.exp(-(x - size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(size)])
    return g / g.sum()


def _device_sums(img1, img2, window_size):
    from .. import _lib
    a, b = img1.contiguous(), img2.contiguous()
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape
            and a.dim() == 4 and a.shape[1] >= 3):
        raise RuntimeError("PSNR/SSIM kernel needs two fp32 CUDA NCHW tensors of equal shape with >= 3 channels")
    n, c, h, w = a.shape
    win = _gaussian(window_size).contiguous()
    sq = torch.empty(n, device=a.device)
    ss = torch.empty(n, device=a.device)
    ws = torch.empty(_lib.load().focr_psnr_ssim_ws_floats(n, h, w), device=a.device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())                      # noqa: E731
    _lib.call("focr_psnr_ssim", p(a), p(b), p(win), window_size, p(sq), p(ss), p(ws), n, c, h, w,
              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return sq, ss, 3 * h * w


def calculate_psnr(img1, img2):
    if img1.is_cuda:
        sq, _, per = _device_sums(img1, img2, 11)
        mse = sq.sum() / (per * img1.shape[0])
    else:
        mse = ((img1[:, :3] * 255 - img2[:, :3] * 255) ** 2).mean()
    if mse == 0:
        return float("inf")
    return 20 * torch.log10(255.0 / torch.sqrt(mse))


def _window(size, sigma, channel, device, dtype):
    g = _gaussian(size, sigma).to(dtype).unsqueeze(1)
    return (g @ g.t()).expand(channel, 1, size, size).contiguous().to(device)


class SSIM(torch.nn.Module):
    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average = window_size, size_average

    def forward(self, img1, img2):
        if img1.is_cuda:
            _, ss, per = _device_sums(img1, img2, self.window_size)
            return ss.sum() / (per * img1.shape[0]) if self.size_average else ss / per
        img1, img2 = img1[:, :3], img2[:, :3]
        c = img1.shape[1]
        w = _window(self.window_size, 1.5, c, img1.device, img1.dtype)
        pad = self.window_size // 2
        blur = lambda t: F.conv2d(t, w, padding=pad, groups=c)         # noqa: E731
        mu1, mu2 = blur(img1), blur(img2)
        s11 = blur(img1 * img1) - mu1 * mu1
        s22 = blur(img2 * img2) - mu2 * mu2
        s12 = blur(img1 * img2) - mu1 * mu2
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
        return m.mean() if self.size_average else m.mean(1).mean(1).mean(1)


def psnr_ssim(img1, img2, window_size=11):
    """both metrics from one device pass (what TextSR.eval uses): returns (psnr, ssim) 0-d tensors"""
    sq, ss, per = _device_sums(img1, img2, window_size)
    n = img1.shape[0]
    mse = sq.sum() / (per * n)
    return 20 * torch.log10(255.0 / torch.sqrt(mse)), ss.sum() / (per * n)

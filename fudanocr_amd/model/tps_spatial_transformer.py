"""TPS rectifier of the SR nets, same class name / buffers as the reference
(scene-text-telescope/model/tps_spatial_transformer.py:54-111); the grid construction and
bilinear sampling run in one HIP kernel (csrc/pool_sample.hip)."""
import torch
from torch import nn

from .. import kernels as K


def _tps_basis(points, ctrl):
    """U(r) = 0.5 r^2 log r^2 between two point sets, 0 where the points coincide
    (reference compute_partial_repr, tps_spatial_transformer.py:22-34)."""
    d = points.unsqueeze(1) - ctrl.unsqueeze(0)
    r2 = (d * d).sum(-1)
    basis = 0.5 * r2 * torch.log(r2)
    return torch.nan_to_num(basis, nan=0.0)


def _frame_points(n_ctrl, margin_x, margin_y):
    """n/2 points along the top margin line then n/2 along the bottom one, (x, y) in [0,1]
    (reference build_output_control_points, :38-50)."""
    half = n_ctrl // 2
    xs = torch.linspace(margin_x, 1.0 - margin_x, half, dtype=torch.float64)
    pts = torch.zeros(n_ctrl, 2, dtype=torch.float64)
    pts[:half, 0], pts[half:, 0] = xs, xs
    pts[:half, 1], pts[half:, 1] = margin_y, 1.0 - margin_y
    return pts.float()


class TPSSpatialTransformer(nn.Module):
    def __init__(self, output_image_size=None, num_control_points=None, margins=None):
        super().__init__()
        self.output_image_size = output_image_size
        self.num_control_points = num_control_points
        self.margins = margins
        self.target_height, self.target_width = output_image_size
        n = num_control_points
        ctrl = _frame_points(n, *margins)
        system = torch.zeros(n + 3, n + 3)
        system[:n, :n] = _tps_basis(ctrl, ctrl)
        system[:n, n] = 1.0
        system[n, :n] = 1.0
        system[:n, n + 1:] = ctrl
        system[n + 1:, :n] = ctrl.t()
        h, w = self.target_height, self.target_width
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32),
                                indexing="ij")
        xy = torch.stack([gx.reshape(-1) / (w - 1), gy.reshape(-1) / (h - 1)], 1)
        rep = torch.cat([_tps_basis(xy, ctrl), torch.ones(h * w, 1), xy], 1)
        self.register_buffer("inverse_kernel", torch.inverse(system).contiguous())
        self.register_buffer("padding_matrix", torch.zeros(3, 2))
        self.register_buffer("target_coordinate_repr", rep.contiguous())
        self.register_buffer("target_control_points", ctrl)

    def forward(self, input, source_control_points):
        """input: NHWC image, source_control_points [B, n, 2] -> (warped NHWC image, None)."""
        assert source_control_points.dim() == 3 and source_control_points.size(1) == self.num_control_points
        out = K.tps_warp(input, source_control_points, self.inverse_kernel, self.target_coordinate_repr)
        return out, None

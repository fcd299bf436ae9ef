"""TSRN (scene-text-telescope BiGRU-based SR net) on HIP kernels; same public classes, constructor
signatures and state_dict keys as the reference (scene-text-telescope/model/tsrn.py:18-145).
I/O NCHW at the module boundary, channel-last inside: the reference's H<->W transposes around
gru1 (tsrn.py:96) disappear -- the GRU kernel scans the NHWC map vertically in place."""
import math

import torch
from torch import nn

from .. import kernels as K
from ._layers import BatchNorm2d, Conv2d, PReLU
from .stn_head import STNHead
from .tbsrn import UpsampleBLock, mish  # noqa: F401  (same definitions in both reference files)
from .tps_spatial_transformer import TPSSpatialTransformer


class GruBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        assert out_channels % 2 == 0 and out_channels == 64, "the path uses 64 channels (hidden 32)"
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=1, padding=0)
        self.gru = nn.GRU(out_channels, out_channels // 2, bidirectional=True, batch_first=True)  # registry
        self._packed_gru = None       # (wih, bih, whh, bhh) views of the engine's flat buffers (TrainStep._attach_packed_gru)

    def forward(self, x, vertical=False):
        """x: NHWC map.  vertical=False: sequences run along W (gru2); True: along H (gru1)."""
        g = self.gru
        b, h, w, c = x.shape
        y = self.conv1(x).view(b * h * w, c)
        if self._packed_gru is not None and self._packed_gru[0].data_ptr() == g.weight_ih_l0.data_ptr():
            wih, bih, whh, bhh = self._packed_gru
        else:
            wih = torch.cat([g.weight_ih_l0, g.weight_ih_l0_reverse], 0)
            bih = torch.cat([g.bias_ih_l0, g.bias_ih_l0_reverse], 0)
            whh = torch.stack([g.weight_hh_l0, g.weight_hh_l0_reverse], 0)
            bhh = torch.stack([g.bias_hh_l0, g.bias_hh_l0_reverse], 0)
        gx = K.linear(y, wih, bih)
        if vertical:
            out = K.gru_recurrence(gx, whh, bhh, b * w, h, w, h * w, 1, w)
        else:
            out = K.gru_recurrence(gx, whh, bhh, b * h, w, 1, w, 0, 1)
        return out.view(b, h, w, c)


class RecurrentResidualBlock(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv1 = Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn1 = BatchNorm2d(channels)
        self.gru1 = GruBlock(channels, channels)
        self.prelu = mish()
        self.conv2 = Conv2d(channels, channels, kernel_size=3, padding=1)
        self.bn2 = BatchNorm2d(channels)
        self.gru2 = GruBlock(channels, channels)

    def forward(self, x):
        r = K.conv_bn(x, self.conv1, self.bn1, act=K.ACT_MISH)
        r = K.conv_bn(r, self.conv2, self.bn2)
        r = self.gru1(r, vertical=True)
        return self.gru2(K.add(x, r))


class TSRN(nn.Module):
    def __init__(self, scale_factor=2, width=128, height=32, STN=False, srb_nums=5, mask=False, hidden_units=32):
        super().__init__()
        in_planes = 4 if mask else 3
        assert math.log(scale_factor, 2) % 1 == 0
        upsample_block_num = int(math.log(scale_factor, 2))
        c = 2 * hidden_units
        self.block1 = nn.Sequential(Conv2d(in_planes, c, kernel_size=9, padding=4), PReLU())
        self.srb_nums = srb_nums
        for i in range(srb_nums):
            setattr(self, "block%d" % (i + 2), RecurrentResidualBlock(c))
        setattr(self, "block%d" % (srb_nums + 2),
                nn.Sequential(Conv2d(c, c, kernel_size=3, padding=1), BatchNorm2d(c)))
        tail = [UpsampleBLock(c, 2) for _ in range(upsample_block_num)]
        tail.append(Conv2d(c, in_planes, kernel_size=9, padding=4))
        setattr(self, "block%d" % (srb_nums + 3), nn.Sequential(*tail))
        self.tps_inputsize = [height // scale_factor, width // scale_factor]
        self.stn = STN
        if self.stn:
            self.tps = TPSSpatialTransformer(output_image_size=tuple(self.tps_inputsize),
                                             num_control_points=20, margins=(0.05, 0.05))
            self.stn_head = STNHead(in_planes=in_planes, num_ctrlpoints=20, activation="none")

    def forward(self, x):
        x = K.to_nhwc(x)
        if self.stn and self.training:
            _, ctrl = self.stn_head(x)
            x, _ = self.tps(x, ctrl)
        b1 = self.block1(x)
        h = b1
        for i in range(self.srb_nums):
            h = getattr(self, "block%d" % (i + 2))(h)
        tail7 = getattr(self, "block%d" % (self.srb_nums + 2))
        h = K.conv_bn(h, tail7[0], tail7[1], residual=b1)
        h = getattr(self, "block%d" % (self.srb_nums + 3))(h)
        return K.to_nchw(h, tanh=True)

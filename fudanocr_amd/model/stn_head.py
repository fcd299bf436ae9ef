"""Control-point regressor of the STN, same class name / state_dict keys as the reference
(scene-text-telescope/model/stn_head.py:25-99); convolutions, BN+ReLU, pooling and the FC
layers run on the HIP kernels in channel-last layout."""
import math

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ._layers import BatchNorm1d, BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLUTag


def conv3x3_block(in_planes, out_planes, stride=1):
    return nn.Sequential(Conv2d(in_planes, out_planes, kernel_size=3, stride=1, padding=1),
                         BatchNorm2d(out_planes), ReLUTag(inplace=True))


class STNHead(nn.Module):
    def __init__(self, in_planes, num_ctrlpoints, activation="none"):
        super().__init__()
        self.in_planes, self.num_ctrlpoints, self.activation = in_planes, num_ctrlpoints, activation
        assert activation == "none", "only the activation the SR nets use is built"
        self.stn_convnet = nn.Sequential(
            conv3x3_block(in_planes, 32), MaxPool2d(kernel_size=2, stride=2),
            conv3x3_block(32, 64), MaxPool2d(kernel_size=2, stride=2),
            conv3x3_block(64, 128), MaxPool2d(kernel_size=2, stride=2),
            conv3x3_block(128, 256), MaxPool2d(kernel_size=2, stride=2),
            conv3x3_block(256, 256), MaxPool2d(kernel_size=(1, 2), stride=(1, 2)),
            conv3x3_block(256, 256))
        self.stn_fc1 = nn.Sequential(Linear(2 * 256, 512), BatchNorm1d(512), ReLUTag(inplace=True))
        self.stn_fc2 = Linear(512, num_ctrlpoints * 2)
        self._init_weights()

    @torch.no_grad()
    def _init_weights(self):
        # stn_head.py:55-86: conv N(0, sqrt(2/(k*k*Cout))), BN (1,0), Linear N(0,1e-3); fc2 starts
        # as the constant frame of control points (weight 0, bias = points on a 0.01 margin)
        for m in list(self.stn_convnet.modules()) + list(self.stn_fc1.modules()):
            if isinstance(m, nn.Conv2d):
                m.weight.normal_(0, math.sqrt(2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)))
                m.bias.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.fill_(1)
                m.bias.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.normal_(0, 0.001)
                m.bias.zero_()
        half = self.num_ctrlpoints // 2
        xs = np.linspace(0.01, 0.99, half)
        frame = np.concatenate([np.stack([xs, np.full(half, 0.01)], 1), np.stack([xs, np.full(half, 0.99)], 1)])
        self.stn_fc2.weight.zero_()
        self.stn_fc2.bias.copy_(torch.from_numpy(frame.astype(np.float32)).reshape(-1))

    def forward(self, x):
        """x: NHWC image -> (img_feat [B,512], control points [B,n,2])."""
        for m in self.stn_convnet:
            if isinstance(m, nn.Sequential):
                # conv -> BatchNorm -> ReLU; where the convolution runs on the halo kernel (the 64 -> 128 layer on 4 x 16 maps)
                # its epilogue hands the BatchNorm the per-tile sums: no statistics passes over the 8192-row output
                x = K.conv_bn(x, m[0], m[1], act=K.ACT_RELU)
            else:
                x = m(x)
        b = x.shape[0]
        feat = K.to_nchw(x).reshape(b, -1)                  # reference flattens NCHW: index c*W + w
        fc, bn = self.stn_fc1[0], self.stn_fc1[1]
        img_feat = bn(fc(feat), act=K.ACT_RELU)
        pts = self.stn_fc2(img_feat, alpha=0.1)              # fc2(0.1 * feat), stn_head.py:93
        return img_feat, pts.view(-1, self.num_ctrlpoints, 2)

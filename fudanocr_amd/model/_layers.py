"""Parameter containers with the reference's attribute names whose forward runs on the HIP
kernels (channel-last activations).  torch.nn classes are used only as parameter/buffer
registries (so `state_dict()` keys, shapes and default initialisers equal the reference's);
none of their forward implementations is ever called.
"""
import torch
from torch import nn

from .. import kernels as K


def _channels_last_(conv):
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    return conv


class Conv2d(nn.Conv2d):
    """NHWC convolution, stride 1 (all convolutions of the path are stride 1)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        assert self.stride == (1, 1) and self.dilation == (1, 1) and self.groups == 1
        _channels_last_(self)

    def forward(self, x, residual=None, relu=False, take_deferred=False):
        return K.conv2d(x, self.weight, self.bias, pad=self.padding, residual=residual, relu=relu,
                        take_deferred=take_deferred)


class Linear(nn.Linear):
    def forward(self, x, residual=None, relu=False, alpha=1.0, dropout=0.0, take_deferred=False, defer_residual=False,
                fuse_input_relu=False):
        return K.linear(x, self.weight, self.bias, residual=residual, relu=relu, alpha=alpha, dropout=dropout,
                        take_deferred=take_deferred, defer_residual=defer_residual, fuse_input_relu=fuse_input_relu)


class _BNMixin:
    def forward(self, x, act=K.ACT_NONE, residual=None):
        training = self.training or not self.track_running_stats
        return K.batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var,
                               self.num_batches_tracked if training else None, training, act=act,
                               residual=residual, momentum=self.momentum, eps=self.eps)


class BatchNorm2d(_BNMixin, nn.BatchNorm2d):
    """BatchNorm over the channel-last axis of an NHWC tensor (+ fused activation / residual)."""


class BatchNorm1d(_BNMixin, nn.BatchNorm1d):
    """BatchNorm over the last axis of a [rows, C] matrix."""


class PReLU(nn.PReLU):
    def forward(self, x):
        return K.prelu(x, self.weight)


class ReLUTag(nn.ReLU):
    """The reference's nn.ReLU module slot.  The product's containers (CRNN.forward, STNHead) fuse this ReLU into the
    preceding convolution / BatchNorm epilogue and skip the module; called directly (e.g. through a plain
    nn.Sequential walk) it IS a ReLU, so no path can silently lose the non-linearity."""

    def forward(self, x):
        return torch.relu(x)


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x, relu_input=False):
        k = self.kernel_size if isinstance(self.kernel_size, tuple) else (self.kernel_size,) * 2
        s = self.stride if isinstance(self.stride, tuple) else (self.stride,) * 2
        p = self.padding if isinstance(self.padding, tuple) else (self.padding,) * 2
        return K.maxpool(x, k, s, p, relu_input=relu_input)

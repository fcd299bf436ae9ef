"""CRNN recognizer on HIP kernels; same classes / state_dict keys as the reference
(scene-text-telescope/model/crnn/crnn.py:6-80).  Input NCHW [B, nc, 32, 100], output logits
[T=26, B, nclass].  The training step uses the recognizer frozen (eval-mode BN, no weight gradients) exactly as
the reference does (interfaces/super_resolution.py:168-171): gradients flow to its input.  With requires_grad left on
it trains like any other module (convolution / BatchNorm / Linear weight gradients as everywhere, LSTM weight
gradients from kernels._LSTMRecur)."""
import os

import torch
from torch import nn

from ... import kernels as K
from .._layers import BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLUTag


# FOCR_CRNN_FOLD_BN=0: frozen conv -> BatchNorm -> relu layers as convolution + eval-BatchNorm passes (A/B switch)
_FOLD_BN = os.environ.get("FOCR_CRNN_FOLD_BN", "1") != "0"


def _pair(v):
    return v if isinstance(v, tuple) else (v, v)


class BidirectionalLSTM(nn.Module):
    def __init__(self, nIn, nHidden, nOut):
        super().__init__()
        self.rnn = nn.LSTM(nIn, nHidden, bidirectional=True)      # parameter registry only
        self.embedding = Linear(nHidden * 2, nOut)
        self._packed = None

    def _pack(self):
        """[W_ih | W_ih_reverse], stacked W_hh / biases, rebuilt only when a parameter changed."""
        r = self.rnn
        ps = (r.weight_ih_l0, r.weight_ih_l0_reverse, r.weight_hh_l0, r.weight_hh_l0_reverse,
              r.bias_ih_l0, r.bias_ih_l0_reverse, r.bias_hh_l0, r.bias_hh_l0_reverse)
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if self._packed is None or self._packed[0] != key:
            with torch.no_grad():
                wih = torch.cat([ps[0], ps[1]], 0).contiguous()
                whh = torch.stack([ps[2], ps[3]], 0).contiguous()
                bih = torch.cat([ps[4], ps[5]], 0).contiguous()
                bhh = torch.stack([ps[6], ps[7]], 0).contiguous()
            self._packed = (key, wih, whh, bih, bhh)
        return self._packed[1:]

    def forward(self, input, t_len=None, batch=None, st_t=None, st_b=None):
        """input: rows x nIn with row(t,b) = t*st_t + b*st_b (defaults: sequence-first [T,B,nIn])."""
        if input.dim() == 3:
            t_len, batch = input.shape[0], input.shape[1]
            st_t, st_b = batch, 1
            input = input.reshape(t_len * batch, -1)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.rnn.parameters()):
            # trainable recognizer (optimise the CRNN together with the SR net): pack through autograd so
            # the gradients of the packed operands flow back to the eight nn.LSTM parameters
            self._packed = None        # an engine may rewrite them through raw pointers (no version bump): the frozen-path
            #                            pack -- and with it kernels._lstm_prepared's split, keyed by that tensor -- is void
            r = self.rnn
            wih = torch.cat([r.weight_ih_l0, r.weight_ih_l0_reverse], 0)
            whh = torch.stack([r.weight_hh_l0, r.weight_hh_l0_reverse], 0)
            bih = torch.cat([r.bias_ih_l0, r.bias_ih_l0_reverse], 0)
            bhh = torch.stack([r.bias_hh_l0, r.bias_hh_l0_reverse], 0)
        else:
            wih, whh, bih, bhh = self._pack()
        gx = K.linear(input, wih, bih)
        rec = K.lstm_recurrence(gx, whh, bhh, t_len, batch, st_t, st_b)     # [T,B,2H]
        out = self.embedding(rec.view(t_len * batch, -1))
        return out.view(t_len, batch, -1)


class CRNN(nn.Module):
    def __init__(self, imgH, nc, nclass, nh, n_rnn=2, leakyRelu=False):
        super().__init__()
        assert imgH % 16 == 0, "imgH has to be a multiple of 16"
        assert not leakyRelu, "the path uses ReLU"
        ks, ps = [3, 3, 3, 3, 3, 3, 2], [1, 1, 1, 1, 1, 1, 0]
        nm = [64, 128, 256, 256, 512, 512, 512]
        cnn = nn.Sequential()
        for i in range(7):
            cnn.add_module("conv%d" % i, Conv2d(nc if i == 0 else nm[i - 1], nm[i], ks[i], 1, ps[i]))
            if i in (2, 4, 6):
                cnn.add_module("batchnorm%d" % i, BatchNorm2d(nm[i]))
            cnn.add_module("relu%d" % i, ReLUTag(True))
            if i in (0, 1):
                cnn.add_module("pooling%d" % i, MaxPool2d(2, 2))
            elif i in (3, 5):
                cnn.add_module("pooling%d" % (2 if i == 3 else 3), MaxPool2d((2, 2), (2, 1), (0, 1)))
        self.cnn = cnn
        self.rnn = nn.Sequential(BidirectionalLSTM(512, nh, nh), BidirectionalLSTM(nh, nh, nclass))

    @staticmethod
    def _foldable(conv, bn):
        """conv -> BatchNorm(eval) -> relu with nothing to train: fold the normalisation into the convolution"""
        if not _FOLD_BN or bn.training or not bn.track_running_stats or not conv.weight.is_cuda:
            return False
        ps = (conv.weight, conv.bias, bn.weight, bn.bias)
        return not any(p is not None and p.requires_grad for p in ps)

    def _folded(self, name, conv, bn):
        """(weight * a[co], (bias - mean) * a + beta) with a = gamma / sqrt(running_var + eps), cached until one of the
        six tensors changes (version counters); the weight keeps the channels_last layout the kernels read"""
        ts = (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts) + (bn.eps,)
        cache = self.__dict__.setdefault("_fold_cache", {})
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                wf = (conv.weight * a.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
                b0 = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
                bf = ((b0 - bn.running_mean) * a + bn.bias).contiguous()
            if hit is not None:
                # keep the replaced tensors alive: a new fold must never reuse their addresses, which key the frozen
                # entries of kernels.FlipTable / FragTable (rebuilds are rare: checkpoint loads, freeze cycles)
                self.__dict__.setdefault("_fold_retired", []).append(hit[1:])
            cache[name] = hit = (key, wf, bf)
        return hit[1], hit[2]

    def forward(self, input):
        x = K.to_nhwc(input) if input.shape[1] != 1 else input.reshape(input.shape[0], input.shape[2],
                                                                       input.shape[3], 1)
        mods = list(self.cnn.named_children())
        i = 0
        after_conv_relu = False
        while i < len(mods):
            name, m = mods[i]
            if name.startswith("conv"):
                bn = mods[i + 1][1] if mods[i + 1][0].startswith("batchnorm") else None
                pool = mods[i + 2][1] if bn is None and i + 2 < len(mods) and mods[i + 2][0].startswith("pooling") else None
                if pool is not None and K.conv0_relu_pool_supported(x, m.weight, m.bias, _pair(pool.kernel_size),
                                                                    _pair(pool.stride), _pair(pool.padding)):
                    # frozen first layer: conv + relu + 2x2 pooling as one launch each way (csrc/crnn_conv0_pool.hip)
                    x = K.conv0_relu_pool(x, m.weight, m.bias)
                    i += 3                      # conv, relu, pooling
                    after_conv_relu = False
                elif bn is None:
                    x = m(x, relu=True)
                    i += 2                      # conv, relu
                    after_conv_relu = True
                elif self._foldable(m, bn):
                    # frozen layer with eval-mode statistics: BatchNorm is a per-channel affine map of the convolution's
                    # output -- folded into its weights and bias once, the layer is conv + relu with no BatchNorm pass at
                    # all (forward: one launch less and no second tensor; backward: no eval-BatchNorm gradient pass)
                    wf, bf = self._folded(name, m, bn)
                    x = K.conv2d(x, wf, bf, pad=m.padding, relu=True)
                    i += 3                      # conv, batchnorm, relu
                    after_conv_relu = False
                else:
                    # trainable layer or train-mode statistics: its tensors may be rewritten behind autograd's version
                    # counters (raw-pointer Adam, BatchNorm running statistics) -- a fold cached earlier is void
                    self.__dict__.setdefault("_fold_cache", {}).pop(name, None)
                    x = bn(m(x), act=K.ACT_RELU)
                    i += 3                      # conv, batchnorm, relu
                    after_conv_relu = False
            else:
                # pooling right behind conv + relu: its backward also applies that relu's backward (kernels._MaxPool)
                x = m(x, relu_input=after_conv_relu) if name.startswith("pooling") else m(x)
                after_conv_relu = False
                i += 1
        b, h, w, c = x.shape
        assert h == 1, "the height of conv must be 1"
        # rows of x are (b, t): feed the LSTM through strides instead of transposing to [T,B,C]
        out = self.rnn[0](x.view(b * w, c), t_len=w, batch=b, st_t=1, st_b=w)
        return self.rnn[1](out)

"""TBSRN (text-gestalt / scene-text-telescope transformer-based SR net) on HIP kernels.

Same public classes, constructor signatures and state_dict keys as the reference
(scene-text-telescope/model/tbsrn.py:23-305; text-gestalt/model/tbsrn.py is identical up to
dead timing lines), including the parameters that never receive a gradient (`conv`, `bn`,
`blockK.gru1/gru2`, `compress_attention_linear`) so reference checkpoints
load.  I/O is NCHW at the module boundary, channel-last inside.
"""
import math
import os

import torch
from torch import nn

from .. import kernels as K
from ._layers import BatchNorm2d, Conv2d, Linear, PReLU
from .stn_head import STNHead
from .tps_spatial_transformer import TPSSpatialTransformer

# FOCR_FE_FUSED=0: the FeatureEnhancer as separate per-layer kernels (A/B measurements; precision mode 0 always does)
_FE_FUSED = os.environ.get("FOCR_FE_FUSED", "1") != "0"
# FOCR_SRB_FUSED=0: one autograd node per layer of a residual block instead of one per block (A/B: host time)
_SRB_FUSED = os.environ.get("FOCR_SRB_FUSED", "1") != "0"


def positionalencoding2d(d_model, height, width):
    """Fixed 2-D sinusoid table [d_model, H, W] (reference tbsrn.py:39-61): the first half of the
    channels encodes the column, the second half the row; sin on even, cos on odd channels."""
    if d_model % 4 != 0:
        raise ValueError("Cannot use sin/cos positional encoding with odd dimension (got dim=%d)" % d_model)
    half = d_model // 2
    freq = torch.exp(torch.arange(0.0, half, 2) * -(math.log(10000.0) / half))
    col = torch.arange(0.0, width).unsqueeze(1) * freq
    row = torch.arange(0.0, height).unsqueeze(1) * freq
    pe = torch.zeros(d_model, height, width)
    pe[0:half:2] = col.sin().t().unsqueeze(1)
    pe[1:half:2] = col.cos().t().unsqueeze(1)
    pe[half::2] = row.sin().t().unsqueeze(2)
    pe[half + 1::2] = row.cos().t().unsqueeze(2)
    return pe


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std_unbiased + eps) + b_2   (reference tbsrn.py:23-36)."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x, residual=None, defer=False):
        return K.layernorm_std(x, self.a_2, self.b_2, residual=residual, eps=self.eps, defer=defer)


class MultiHeadedAttention(nn.Module):
    def __init__(self, h, d_model, dropout=0.1, compress_attention=False):
        super().__init__()
        assert d_model % h == 0
        self.d_k, self.h = d_model // h, h
        self.linears = nn.ModuleList([Linear(d_model, d_model) for _ in range(4)])
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)          # p and train/eval flag only
        self._packed_qkv = None                        # (weight, bias) views set by the training engine
        self.compress_attention = compress_attention
        self.compress_attention_linear = nn.Linear(h, 1)   # dead in the reference too (tbsrn.py:107)

    def forward(self, query, key, value, mask=None, align=None, take_deferred=False):
        assert mask is None, "the SR nets never pass a mask"
        p = self.dropout.p if self.dropout.training else 0.0
        if query is key and key is value:
            # self-attention (the only use in the SR nets): one packed [rows, 3*d] projection -- the tokens are
            # read once, and the backward is one dgrad GEMM instead of three plus two gradient adds
            if self._packed_qkv is not None and self._packed_qkv[0].data_ptr() == self.linears[0].weight.data_ptr():
                w, b = self._packed_qkv       # views of the engine's flat buffers (engine.TrainStep._attach_packed_qkv)
            else:
                w = torch.cat([self.linears[0].weight, self.linears[1].weight, self.linears[2].weight], 0)
                b = torch.cat([self.linears[0].bias, self.linears[1].bias, self.linears[2].bias], 0)
            ctx = K.attention_packed(K.linear(query, w, b, take_deferred=take_deferred), heads=self.h, p_drop=p)
        else:
            q, k, v = (lin(x) for lin, x in zip(self.linears, (query, key, value)))
            ctx = K.attention(q, k, v, heads=self.h, p_drop=p)
        return self.linears[3](ctx), None


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = Linear(d_model, d_ff)
        self.w_2 = Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, take_deferred=False):
        # Dropout(relu(w_1 x)) in one kernel: the dropout is part of the GEMM epilogue (and of its relu backward)
        p = self.dropout.p if self.dropout.training else 0.0
        # h has exactly one consumer (w_2): its relu-dropout backward rides in w_2's data-gradient epilogue
        return self.w_2(self.w_1(x, relu=True, dropout=p, take_deferred=take_deferred), fuse_input_relu=True)


class FeatureEnhancer(nn.Module):
    def __init__(self):
        super().__init__()
        self.multihead = MultiHeadedAttention(h=4, d_model=128, dropout=0.1)
        self.mul_layernorm1 = LayerNorm(features=128)
        self.pff = PositionwiseFeedForward(128, 128)
        self.mul_layernorm3 = LayerNorm(features=128)
        self.linear = Linear(128, 64)
        self._pe = None

    def _pe_table(self, device):
        # the reference rebuilds this on the host and copies it every call (tbsrn.py:83)
        if self._pe is None or self._pe.device != device:
            self._pe = positionalencoding2d(64, 16, 64).reshape(64, 1024).t().contiguous().to(device)
        return self._pe

    def _fused_params(self):
        """the 14 tensors of kernels.FE_PARAM_NAMES (packed q | k | v projection first)"""
        mh, ln1, ln3, pff = self.multihead, self.mul_layernorm1, self.mul_layernorm3, self.pff
        assert ln1.eps == ln3.eps and mh.h == 4
        if mh._packed_qkv is not None and mh._packed_qkv[0].data_ptr() == mh.linears[0].weight.data_ptr():
            wqkv, bqkv = mh._packed_qkv       # views of the engine's flat buffers
        else:
            wqkv = torch.cat([mh.linears[0].weight, mh.linears[1].weight, mh.linears[2].weight], 0)
            bqkv = torch.cat([mh.linears[0].bias, mh.linears[1].bias, mh.linears[2].bias], 0)
        return (wqkv, bqkv, mh.linears[3].weight, mh.linears[3].bias, ln1.a_2, ln1.b_2, pff.w_1.weight, pff.w_1.bias,
                pff.w_2.weight, pff.w_2.bias, ln3.a_2, ln3.b_2, self.linear.weight, self.linear.bias)

    def _fused_dropout(self):
        mh, pff = self.multihead, self.pff
        return (mh.dropout.p if mh.dropout.training else 0.0, pff.dropout.p if pff.dropout.training else 0.0)

    def forward(self, conv_feature, residual=None, defer_block_input=False):
        """conv_feature: [B, 1024, 64] tokens (channel-last) -> [B, 1024, 64] (+ residual)."""
        # Each of tok / r / the block input feeds a GEMM AND a later residual slot: the residual consumer parks its
        # gradient (defer) and the GEMM's data-gradient kernel adds it in its epilogue (take_deferred) -- see
        # kernels.py "Deferred residual gradients"; three 67 MB gradient-add passes per block disappear.
        g = torch.is_grad_enabled() and conv_feature.requires_grad
        if _FE_FUSED and K.fe_chain_supported(conv_feature, self.multihead.h, self.multihead.h * self.multihead.d_k):
            # one autograd node for the whole block: the row-local layers run as fused chains (csrc/fe_chain.hip)
            p_attn, p_ffn = self._fused_dropout()
            return K.feature_enhancer_fused(
                conv_feature, residual, self._pe_table(conv_feature.device), self._fused_params(), heads=self.multihead.h,
                p_attn=p_attn, p_ffn=p_ffn, eps=self.mul_layernorm1.eps,
                defer_residual=g and residual is not None and defer_block_input)
        tok = K.concat_pe(conv_feature, self._pe_table(conv_feature.device))
        att, _ = self.multihead(tok, tok, tok, mask=None, take_deferred=g)
        r = self.mul_layernorm1(att, residual=tok, defer=g)
        r = self.mul_layernorm3(self.pff(r, take_deferred=g), residual=r, defer=g)
        return self.linear(r, residual=residual, defer_residual=g and residual is not None and defer_block_input)


class mish(nn.Module):
    """x * tanh(softplus(x)) (reference tbsrn.py:258-266).  The blocks fuse it into the BatchNorm / pixel-shuffle
    kernels; the module's own forward is the same function for direct callers."""

    def __init__(self):
        super().__init__()
        self.activated = True

    def forward(self, x):
        if self.activated:
            x = x * torch.tanh(torch.nn.functional.softplus(x))
        return x


class GruBlock(nn.Module):
    """Parameter holder only: TBSRN constructs but never calls it (tbsrn.py:234,239)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, padding=0)
        self.gru = nn.GRU(out_channels, out_channels // 2, bidirectional=True, batch_first=True)


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
        self.feature_enhancer = FeatureEnhancer()
        for p in self.parameters():                      # tbsrn.py:242-244
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, x):
        g = torch.is_grad_enabled() and x.requires_grad
        mh = self.feature_enhancer.multihead
        if (_SRB_FUSED and _FE_FUSED and mh.h == 4 and mh.h * mh.d_k == 128
                and K.srb_fused_supported(x, self.conv1, self.conv2, self.bn1, self.bn2)):
            # training mode: the whole block is one autograd node (kernels._SRBFused), same library calls
            fe = self.feature_enhancer
            p_attn, p_ffn = fe._fused_dropout()
            return K.srb_fused(x, fe._pe_table(x.device), self.conv1, self.bn1, self.conv2, self.bn2, fe._fused_params(),
                               heads=fe.multihead.h, p_attn=p_attn, p_ffn=p_ffn, eps_ln=fe.mul_layernorm1.eps)
        r = K.conv_bn(x, self.conv1, self.bn1, act=K.ACT_MISH, take_deferred=g)
        r = K.conv_bn(r, self.conv2, self.bn2)
        n, h, w, c = r.shape
        out = self.feature_enhancer(r.view(n, h * w, c), residual=x.view(n, h * w, c), defer_block_input=g)
        return out.view(n, h, w, c)


class UpsampleBLock(nn.Module):
    def __init__(self, in_channels, up_scale):
        super().__init__()
        assert up_scale == 2
        self.conv = Conv2d(in_channels, in_channels * up_scale ** 2, kernel_size=3, padding=1)
        self.pixel_shuffle = nn.PixelShuffle(up_scale)
        self.prelu = mish()

    def forward(self, x):
        return K.pixelshuffle_mish(self.conv(x))


class TBSRN(nn.Module):
    def __init__(self, scale_factor=2, width=128, height=32, STN=True, srb_nums=5, mask=False,
                 hidden_units=32, input_channel=3):
        super().__init__()
        self.conv = nn.Conv2d(input_channel, 3, 3, 1, 1)        # dead (tbsrn.py:170-172)
        self.bn = nn.BatchNorm2d(3)
        self.relu = nn.ReLU()
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
        """x: [B, Cin, 16, 64] NCHW in [0,1] -> SR image [B, Cin, 32, 128] NCHW in (-1,1)."""
        K.check_deferred()             # a parked residual gradient of an earlier backward must have been consumed
        x = K.to_nhwc(x)
        if self.stn and self.training:
            _, ctrl = self.stn_head(x)
            x, _ = self.tps(x, ctrl)
        b1 = self.block1(x)
        h = b1
        for i in range(self.srb_nums):
            h = getattr(self, "block%d" % (i + 2))(h)
        tail7 = getattr(self, "block%d" % (self.srb_nums + 2))
        h = K.conv_bn(h, tail7[0], tail7[1], residual=b1)      # block1 + block7
        h = getattr(self, "block%d" % (self.srb_nums + 3))(h)
        return K.to_nchw(h, tanh=True)

"""Stroke-level-decomposition transformer recognizer on the HIP kernels: same classes, constructor signatures, forward
arguments / result dict and `state_dict` keys as the reference's model/transformer.py:19-377 (ResNet-[3,4,6,3] encoder
to 1024 channels on 16 x 16 maps, one decoder block with masked self-attention + cross-attention over the 256 image
positions, 4 heads x 256, generator over the 7 stroke classes).

Activations are channel-last inside the encoder (NHWC; its 3x3 convolutions run on the halo kernel with BatchNorm
statistics from the conv epilogue), so the decoder's memory `conv_feature.view(b,c,h*w).permute(0,2,1)` is the encoder
output itself -- no transpose.  torch.nn classes are parameter registries only (reference key names and order)."""
import math

import torch
import torch.nn as nn

from ... import kernels as K
from ...model._layers import BatchNorm2d, Conv2d, Linear, MaxPool2d, ReLUTag
from .. import ops
from ..util import get_alphabet


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, downsample):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, kernel_size=3, stride=1, padding=1)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLUTag(inplace=True)
        self.conv2 = Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        if K.conv_bn_foldable(self.conv2, self.bn2, x) and K.conv_bn_foldable(self.conv1, self.bn1, x):
            # frozen, eval-mode block (the recognizers inside the text- / stroke-focus losses): relu(bn2(conv2) + residual)
            # is one convolution launch on folded weights.  Backward (only data gradients exist): with an identity
            # shortcut every gradient of x arrives through conv1 -- the shortcut's is parked by conv2 and added in conv1's
            # data-gradient epilogue -- so that epilogue can also apply the relu backward of the layer that produced x; `out`
            # has conv2 as its only consumer, the same there: two relu-backward launches and one gradient add per block gone
            ident = self.downsample is None
            out = K.conv_bn(x, self.conv1, self.bn1, act=K.ACT_RELU, take_deferred=ident, fuse_input_relu=ident)
            residual = x if ident else K.conv_bn(x, self.downsample[0], self.downsample[1])
            return K.conv_bn(out, self.conv2, self.bn2, residual=residual, relu_out=True, defer_residual=ident,
                             fuse_input_relu=True)
        out = K.conv_bn(x, self.conv1, self.bn1, act=K.ACT_RELU)
        residual = x if self.downsample is None else K.conv_bn(x, self.downsample[0], self.downsample[1])
        if K.conv_bn_foldable(self.conv2, self.bn2, out):
            return K.conv_bn(out, self.conv2, self.bn2, residual=residual, relu_out=True)
        out = K.conv_bn(out, self.conv2, self.bn2)
        return ops.add_relu(out, residual)


class ResNet(nn.Module):
    def __init__(self, num_in, block, layers, pool_before_layer1=False):
        """pool_before_layer1: the text-focus recognizer's ResNet (scene-text-telescope/loss/transformer.py:141) applies
        `layer1_pool`; the stroke-level-decomposition one constructs it but never calls it"""
        super().__init__()
        self.pool_before_layer1 = bool(pool_before_layer1)
        self.conv1 = Conv2d(num_in, 64, kernel_size=3, stride=1, padding=1)
        self.bn1 = BatchNorm2d(64)
        self.relu1 = ReLUTag(inplace=True)
        self.pool = MaxPool2d((2, 2), (2, 2))
        self.conv2 = Conv2d(64, 128, kernel_size=3, stride=1, padding=1)
        self.bn2 = BatchNorm2d(128)
        self.relu2 = ReLUTag(inplace=True)
        self.layer1_pool = MaxPool2d((2, 2), (2, 2))            # constructed, unused (transformer.py:140,147,...)
        self.layer1 = self._make_layer(block, 128, 256, layers[0])
        self.layer1_conv = Conv2d(256, 256, 3, 1, 1)
        self.layer1_bn = BatchNorm2d(256)
        self.layer1_relu = ReLUTag(inplace=True)
        self.layer2_pool = MaxPool2d((2, 2), (2, 2))
        self.layer2 = self._make_layer(block, 256, 256, layers[1])
        self.layer2_conv = Conv2d(256, 256, 3, 1, 1)
        self.layer2_bn = BatchNorm2d(256)
        self.layer2_relu = ReLUTag(inplace=True)
        self.layer3_pool = MaxPool2d((2, 2), (2, 2))
        self.layer3 = self._make_layer(block, 256, 512, layers[2])
        self.layer3_conv = Conv2d(512, 512, 3, 1, 1)
        self.layer3_bn = BatchNorm2d(512)
        self.layer3_relu = ReLUTag(inplace=True)
        self.layer4_pool = MaxPool2d((2, 2), (2, 2))
        self.layer4 = self._make_layer(block, 512, 512, layers[3])
        self.layer4_conv2 = Conv2d(512, 1024, 3, 1, 1)
        self.layer4_conv2_bn = BatchNorm2d(1024)
        self.layer4_conv2_relu = ReLUTag(inplace=True)

    def _make_layer(self, block, inplanes, planes, blocks):
        downsample = None
        if inplanes != planes:
            downsample = nn.Sequential(Conv2d(inplanes, planes, 3, 1, 1), BatchNorm2d(planes))
        layers = [block(inplanes, planes, downsample)]
        for _ in range(1, blocks):
            layers.append(block(planes, planes, downsample=None))
        return nn.Sequential(*layers)

    def forward(self, x):
        """x NHWC [B,32,32,3] -> NHWC [B,16,16,1024]"""
        x = self.pool(K.conv_bn(x, self.conv1, self.bn1, act=K.ACT_RELU))
        x = K.conv_bn(x, self.conv2, self.bn2, act=K.ACT_RELU)
        if self.pool_before_layer1:
            x = self.layer1_pool(x)
        for layer, conv, bn in ((self.layer1, self.layer1_conv, self.layer1_bn),
                                (self.layer2, self.layer2_conv, self.layer2_bn),
                                (self.layer3, self.layer3_conv, self.layer3_bn),
                                (self.layer4, self.layer4_conv2, self.layer4_conv2_bn)):
            x = layer(x)
            x = K.conv_bn(x, conv, bn, act=K.ACT_RELU)
        return x


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout, max_len=7000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        """x: zeros of the embedding's shape in the reference (transformer.py:343): returns dropout(pe[:L]) per sample"""
        b, length, d = x.shape
        pos = self.pe[0, :length].unsqueeze(0).expand(b, length, d).contiguous()
        return K.dropout(pos, self.dropout.p, self.dropout.training)


class MultiHeadedAttention(nn.Module):
    def __init__(self, h, d_model, dropout=0.1, compress_attention=False):
        super().__init__()
        assert d_model % h == 0
        self.d_k, self.h = d_model // h, h
        self.linears = nn.ModuleList([Linear(d_model, d_model) for _ in range(4)])
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)
        self.compress_attention = compress_attention
        self.compress_attention_linear = nn.Linear(h, 1)          # dead parameters (never used by the reference either)

    def forward(self, query, key, value, mask=None, align=None):
        """mask: None or 'causal' (the decoder only ever passes subsequent_mask)"""
        q, k, v = self.linears[0](query), self.linears[1](key), self.linears[2](value)
        p = self.dropout.p if self.dropout.training else 0.0
        x, amap = ops.small_attention(q, k, v, self.h, causal=mask is not None, p_drop=p)
        return self.linears[3](x), amap


class LayerNorm(nn.Module):
    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a = nn.Parameter(torch.ones(features))
        self.b = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x, residual=None):
        return K.layernorm_std(x, self.a, self.b, residual=residual, eps=self.eps)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = Linear(d_model, d_ff)
        self.w_2 = Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        h = self.w_1(x, relu=True)
        return self.w_2(K.dropout(h, self.dropout.p, self.dropout.training))


class Generator(nn.Module):
    def __init__(self, d_model, vocab):
        super().__init__()
        self.proj = Linear(d_model, vocab)
        self.relu = ReLUTag()

    def forward(self, x):
        return self.proj(x)


class Embeddings(nn.Module):
    def __init__(self, d_model, vocab):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        return ops.embedding(x, self.lut.weight, math.sqrt(self.d_model))


class Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.mask_multihead = MultiHeadedAttention(h=4, d_model=1024, dropout=0.1)
        self.mul_layernorm1 = LayerNorm(features=1024)
        self.multihead = MultiHeadedAttention(h=4, d_model=1024, dropout=0.1, compress_attention=True)
        self.mul_layernorm2 = LayerNorm(features=1024)
        self.pff = PositionwiseFeedForward(1024, 2048)
        self.mul_layernorm3 = LayerNorm(features=1024)

    def forward(self, text, conv_feature):
        """text [B,L,1024]; conv_feature NHWC [B,16,16,1024] (its flattened view IS the [B, HW, C] memory)"""
        result = self.mul_layernorm1(self.mask_multihead(text, text, text, mask="causal")[0], residual=text)
        b, hh, ww, c = conv_feature.shape
        mem = conv_feature.view(b, hh * ww, c)
        align, attention_map = self.multihead(result, mem, mem, mask=None)
        result = self.mul_layernorm2(align, residual=result)
        result = self.mul_layernorm3(self.pff(result), residual=result)
        return result, attention_map


class Transformer(nn.Module):
    def __init__(self, mode):
        super().__init__()
        self.mode = mode
        self.word_n_class = len(get_alphabet(mode))
        self.embedding_word = Embeddings(512, self.word_n_class)
        self.pe = PositionalEncoding(d_model=512, dropout=0.1, max_len=7000)
        self.encoder = ResNet(num_in=3, block=BasicBlock, layers=[3, 4, 6, 3])
        self.decoder = Decoder()
        self.generator_word = Generator(1024, self.word_n_class)
        self.attribute = None
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, image, text_length, text_input, conv_feature=None, test=False):
        """image NCHW [B,3,32,32] in [-1,1]; conv_feature (optional, test-time reuse): what a previous call returned
        under 'conv' (NCHW).  Result dict as in the reference: 'pred' ([sum L, 7] ragged in training, [B,L,7] with
        test=True), 'map', 'conv'."""
        if conv_feature is None:
            feat = self.encoder(K.to_nhwc(image))
        else:
            feat = K.to_nhwc(conv_feature)
        if text_length is None:
            return {"conv": K.to_nchw(feat)}
        emb = self.embedding_word(text_input)
        pos = self.pe(emb)
        b, length, _ = emb.shape
        # [emb | pos]: the positional half differs per sample once its dropout is on -> one "period" of B*L rows
        x = K.concat_pe(emb.reshape(1, b * length, -1), pos.reshape(b * length, -1)).view(b, length, -1)
        x, attention_map = self.decoder(x, feat)
        logits = self.generator_word(x)                                    # [B, L, 7]
        if test:
            return {"pred": logits, "map": attention_map, "conv": K.to_nchw(feat)}
        idx = getattr(text_length, "_focr_idx", None)          # device-resident row index (sld/util.py converter)
        if idx is None:
            lens = getattr(text_length, "_focr_host", None)
            if lens is None:
                lens = [int(v) for v in text_length.tolist()]
            idx = torch.tensor([i * length + j for i, n in enumerate(lens) for j in range(n)],
                               dtype=torch.long).to(logits.device, non_blocking=True)
        probs_res = ops.gather_rows(logits.view(b * length, -1), idx)
        return {"pred": probs_res, "map": attention_map, "conv": K.to_nchw(feat)}

"""The recognizer inside the text-focus loss (reference scene-text-telescope/loss/transformer.py:82-389) on the HIP
kernels: ResNet-[1,2,5,3] encoder (1 input channel, two max-pools: 32x128 -> 8x32 maps, 1024 channels), one decoder
block with 16 heads x 64, generator over the 37-symbol alphabet.  Same class names, constructor signatures, forward
signature / return tuple and `state_dict` keys as the reference; it is a sibling of the stroke-level-decomposition
transformer, so the building blocks are shared with fudanocr_amd/sld/model/transformer.py (which cites the twin lines).
The loss uses it frozen and in eval mode (text_focus_loss.py:54-60)."""
import math

import torch
import torch.nn as nn

from .. import kernels as K
from ..model.tbsrn import LayerNorm                      # a_2 / b_2 parameters, unbiased std, eps on std (:226-239)
from ..sld import ops
from ..sld.model.transformer import (BasicBlock, Embeddings, Generator, MultiHeadedAttention,  # noqa: F401
                                     PositionwiseFeedForward, ResNet)

alphabet = "-0123456789abcdefghijklmnopqrstuvwxyz"


def get_alphabet_len():
    return len(alphabet)


class PositionalEncoding(nn.Module):
    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        b, length, d = x.shape
        pos = self.pe[0, :length].unsqueeze(0).expand(b, length, d).contiguous()
        return K.dropout(pos, self.dropout.p, self.dropout.training)


class Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.mask_multihead = MultiHeadedAttention(h=16, d_model=1024, dropout=0.1)
        self.mul_layernorm1 = LayerNorm(features=1024)
        self.multihead = MultiHeadedAttention(h=16, d_model=1024, dropout=0.1, compress_attention=True)
        self.mul_layernorm2 = LayerNorm(features=1024)
        self.pff = PositionwiseFeedForward(1024, 2048)
        self.mul_layernorm3 = LayerNorm(features=1024)

    def forward(self, text, conv_feature, attention_map=None):
        """conv_feature NHWC [B,8,32,1024]: its flattened view is the reference's [B, HW, C] memory"""
        if attention_map is not None:
            raise NotImplementedError("attention_map injection is never used by the text-focus loss")
        r = K.layernorm_std(self.mask_multihead(text, text, text, mask="causal")[0], self.mul_layernorm1.a_2,
                            self.mul_layernorm1.b_2, residual=text, eps=self.mul_layernorm1.eps)
        b, hh, ww, c = conv_feature.shape
        mem = conv_feature.view(b, hh * ww, c)
        align, amap = self.multihead(r, mem, mem, mask=None)
        r = K.layernorm_std(align, self.mul_layernorm2.a_2, self.mul_layernorm2.b_2, residual=r,
                            eps=self.mul_layernorm2.eps)
        r = K.layernorm_std(self.pff(r), self.mul_layernorm3.a_2, self.mul_layernorm3.b_2, residual=r,
                            eps=self.mul_layernorm3.eps)
        return r, amap


class Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.cnn = ResNet(num_in=1, block=BasicBlock, layers=[1, 2, 5, 3], pool_before_layer1=True)

    def forward(self, input):
        """input NCHW [B,1,32,128] -> NHWC [B,8,32,1024]"""
        b, c, h, w = input.shape
        x = input.reshape(b, h, w, 1) if c == 1 else K.to_nhwc(input)
        return self.cnn(x)


class Transformer(nn.Module):
    def __init__(self):
        super().__init__()
        word_n_class = get_alphabet_len()
        self.embedding_word = Embeddings(512, word_n_class)
        self.pe = PositionalEncoding(d_model=512, dropout=0.1, max_len=5000)
        self.encoder = Encoder()
        self.decoder = Decoder()
        self.generator_word = Generator(1024, word_n_class)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_padded(self, image, text_input, attention_map=None):
        """the network on a padded teacher-forcing matrix -> (logits [B, L, 37], word_attention_map [B,16,L,256]); no
        label-dependent shapes, nothing on the host (loss/padded_labels.py)"""
        conv_feature = self.encoder(image)
        emb = self.embedding_word(text_input)
        pos = self.pe(emb)
        b, length, _ = emb.shape
        x = K.concat_pe(emb.reshape(1, b * length, -1), pos.reshape(b * length, -1)).view(b, length, -1)
        x, word_attention_map = self.decoder(x, conv_feature, attention_map=attention_map)
        return self.generator_word(x), word_attention_map

    def forward(self, image, text_length, text_input, test=False, attention_map=None):
        """-> (probs_res [sum L, 37], word_attention_map [B,16,L,256], None); test=True: the padded logits"""
        logits, word_attention_map = self.forward_padded(image, text_input, attention_map=attention_map)
        if test:
            return logits
        b, length = logits.shape[0], logits.shape[1]
        lens = getattr(text_length, "_focr_host", None)
        if lens is None:
            lens = [int(v) for v in text_length.tolist()]
        idx = torch.tensor([i * length + j for i, n in enumerate(lens) for j in range(n)], dtype=torch.long)
        probs_res = ops.gather_rows(logits.view(b * length, -1), idx.to(logits.device, non_blocking=True))
        return probs_res, word_attention_map, None

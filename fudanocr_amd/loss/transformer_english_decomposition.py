"""The stroke-level recognizer inside text-gestalt's stroke-focus loss (reference
text-gestalt/loss/transformer_english_decomposition.py:8,336-398) on the HIP kernels.

Same network as the text-focus recognizer (loss/transformer.py: ResNet-[1,2,5,3] encoder, one 16 x 64 decoder block),
but it reads STROKE sequences: the alphabet is the ten stroke classes `0123456789` (`0` closes a word), the embedding /
generator are registered as `embedding_word_with_upperword` / `generator_word_with_upperword` (the `state_dict` keys of
pretrain_transformer_stroke_decomposition.pth), a 4-channel (masked) image is reduced to its luma inside forward, and the
training branch also returns `correct_list`: per sample, whether the greedy prediction reproduces the teacher-forcing
input (reference :386-394; the stroke-focus loss builds it for HR and SR and, with its `correct_flag = False`, never
uses it)."""
import torch
import torch.nn as nn

from .. import kernels as K
from ..sld import ops
from .transformer import Decoder, Embeddings, Encoder, Generator, PositionalEncoding

alphabet = "0123456789"          # stroke-level alphabet (reference :7-8)


def get_alphabet_len():
    return len(alphabet)


class Transformer(nn.Module):
    def __init__(self):
        super().__init__()
        word_n_class = get_alphabet_len()
        self.embedding_word_with_upperword = Embeddings(512, word_n_class)
        self.pe = PositionalEncoding(d_model=512, dropout=0.1, max_len=5000)
        self.encoder = Encoder()
        self.decoder = Decoder()
        self.generator_word_with_upperword = Generator(1024, word_n_class)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward_padded(self, image, text_input, attention_map=None):
        """the network on a padded teacher-forcing matrix -> (logits [B, L, 10], word_attention_map [B,16,L,256]); no
        label-dependent shapes, nothing on the host (loss/padded_labels.py)"""
        if image.shape[1] == 4:                       # reference :363-367: the mask channel is dropped, RGB -> luma
            image = K.bicubic_gray(image, image.shape[3])
        conv_feature = self.encoder(image)
        emb = self.embedding_word_with_upperword(text_input)
        pos = self.pe(emb)
        b, length, _ = emb.shape
        x = K.concat_pe(emb.reshape(1, b * length, -1), pos.reshape(b * length, -1)).view(b, length, -1)
        x, word_attention_map = self.decoder(x, conv_feature, attention_map=attention_map)
        return self.generator_word_with_upperword(x), word_attention_map

    def forward(self, image, text_length, text_input, test=False, attention_map=None, want_correct=True):
        """-> (probs_res [sum L, 10], word_attention_map [B,16,L,256], correct_list); test=True: the padded logits.
        want_correct=False (used by StrokeFocusLoss, whose correct_flag is off) skips the host read-back of
        correct_list and returns None in its place: no device synchronisation inside the training step."""
        logits, word_attention_map = self.forward_padded(image, text_input, attention_map=attention_map)
        b, length = logits.shape[0], logits.shape[1]
        if test:
            return logits
        lens = getattr(text_length, "_focr_host", None)
        if lens is None:
            lens = [int(v) for v in text_length.tolist()]
        idx = torch.tensor([i * length + j for i, n in enumerate(lens) for j in range(n)], dtype=torch.long)
        probs_res = ops.gather_rows(logits.view(b * length, -1), idx.to(logits.device, non_blocking=True))
        correct_list = None
        if want_correct:
            # greedy class of position j must equal the teacher-forcing input of position j + 1, for j < L - 1
            with torch.no_grad():
                arg = logits.detach().argmax(-1)                                   # [B, Lmax]
                ok = (arg[:, :-1] == text_input[:, 1:]) if length > 1 else arg[:, :0].bool()
                pos_ = torch.arange(max(length - 1, 0), device=arg.device)[None, :]
                valid = pos_ < (torch.as_tensor(lens, device=arg.device)[:, None] - 1)
                correct_list = [bool(v) for v in (ok | ~valid).all(1).tolist()]
        return probs_res, word_attention_map, correct_list

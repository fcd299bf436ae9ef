"""Label codec of the CRNN leg, API of the reference's utils/utils_crnn.py:10-89
(blank = 0, '0-9a-z' -> 1..36, case-insensitive); `collections.Iterable` fixed for py>=3.10."""
import collections.abc

import torch


class strLabelConverter(object):
    def __init__(self, alphabet, ignore_case=True):
        self._ignore_case = ignore_case
        if ignore_case:
            alphabet = alphabet.lower()
        self.alphabet = alphabet + "-"
        self.dict = {ch: i + 1 for i, ch in enumerate(alphabet)}

    def encode(self, text):
        if isinstance(text, str):
            codes = [self.dict[ch.lower() if self._ignore_case else ch] for ch in text]
            lengths = [len(codes)]
        elif isinstance(text, collections.abc.Iterable):
            text = list(text)
            lengths = [len(s) for s in text]
            codes, _ = self.encode("".join(text))
            return codes, torch.IntTensor(lengths)
        return torch.IntTensor(codes), torch.IntTensor(lengths)

    def decode(self, t, length, raw=False):
        if length.numel() == 1:
            n = int(length.reshape(-1)[0])
            assert t.numel() == n, "text with length: {} does not match declared length: {}".format(t.numel(), n)
            if raw:
                return "".join(self.alphabet[i - 1] for i in t.tolist())
            out, prev = [], 0
            for i in t.tolist():
                if i != 0 and i != prev:
                    out.append(self.alphabet[i - 1])
                prev = i
            return "".join(out)
        assert t.numel() == int(length.sum())
        texts, pos = [], 0
        for n in length.tolist():
            texts.append(self.decode(t[pos:pos + n], torch.IntTensor([n]), raw=raw))
            pos += n
        return texts


def get_crnn_pred(outputs):
    """Greedy CTC decode of [B, T, 37] scores (reference interfaces/super_resolution.py:143-158);
    argmax on the device, string assembly on the host."""
    alphabet = "-0123456789abcdefghijklmnopqrstuvwxyz"
    idx = outputs.argmax(2).tolist()
    res = []
    for row in idx:
        s, last = "", 0
        for i in row:
            if i != 0 and i != last:
                s += alphabet[i]
            last = i
        res.append(s)
    return res

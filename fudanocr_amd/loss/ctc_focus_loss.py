"""Training criterion of the measured step: pixel MSE
(reference loss/text_focus_loss.py:44,86) plus, when a recognizer is given, the CTC loss of the
frozen CRNN on the SR output (parse_crnn_data -> CRNN -> log_softmax -> CTC).  Returns the
reference's 4-tuple shape (loss, mse, attention_loss, recognition_loss) with -1 sentinels
(text_focus_loss.py:100-104) -- here `recognition_loss` carries the CTC term."""
import torch
from torch import nn

from .. import kernels as K
from ..utils.utils_crnn import strLabelConverter


class CTCFocusLoss(nn.Module):
    def __init__(self, recognizer=None, alphabet="0123456789abcdefghijklmnopqrstuvwxyz"):
        super().__init__()
        self.recognizer = [recognizer]          # not registered: stays out of state_dict/parameters
        self.converter = strLabelConverter(alphabet)

    MAX_LABEL_LEN = 31      # csrc/ctc.hip: one lane per extended-label position, 2L+1 <= 64

    def encode(self, label_strs, device):
        t, l = self.converter.encode(list(label_strs))
        if int(l.max()) > self.MAX_LABEL_LEN:
            raise ValueError("CTC label longer than %d characters (the CTC kernel's lattice is one 64-lane wave); "
                             "filter such samples in the data pipeline (reference max_len semantics, "
                             "dataset/dataset.py:107-131)" % self.MAX_LABEL_LEN)
        return t.to(device), l.to(device)

    def forward(self, sr_img, hr_img, label_strs=None, encoded=None):
        mse = K.mse_loss(sr_img, hr_img)
        rec = self.recognizer[0]
        if rec is None:
            return mse, mse, -1, -1
        if encoded is None:
            encoded = self.encode(label_strs, sr_img.device)
        gray = K.bicubic_gray(sr_img, 100)                  # parse_crnn_data, base.py:319-325
        logits = rec(gray)                                  # [26, B, 37]
        ctc = K.ctc_loss(logits, encoded[0], encoded[1])
        return mse + ctc, mse, -1, ctc

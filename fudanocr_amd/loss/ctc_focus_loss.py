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

    REPLAY_SAFE = True              # forward(sr, hr, None, encoded) launches kernels only: engine.TrainStep may record it
    LOSS_IS_MSE_PLUS_REC = True     # forward returns loss = mse + ctc exactly: engine.TrainStep may feed d(loss*100) = 100 into both
    MAX_LABEL_LEN = 31      # csrc/ctc.hip: one lane per extended-label position, 2L+1 <= 64
    T_STEPS = 26            # CRNN output length for 32 x 100 inputs (crnn.py:65-80)
    _warned = False

    def encode(self, label_strs, device):
        """Labels longer than the CRNN's 26 output steps have no CTC alignment: F.ctc_loss(zero_infinity=True) gives
        them loss 0 and gradient 0, and so does csrc/ctc.hip (which handles any length as infeasible beyond its
        31-character lattice).  The reference's loaders do not filter them either (dataset.py:87,130: the length test
        never fires), so one long TextZoom label must not abort a run: warn once, keep going."""
        t, l = self.converter.encode(list(label_strs))
        if not CTCFocusLoss._warned and int(l.max()) > self.T_STEPS:
            import warnings
            warnings.warn("CTC: %d label(s) longer than the recognizer's %d output steps contribute loss 0 / gradient 0 "
                          "(as F.ctc_loss with zero_infinity=True)" % (int((l > self.T_STEPS).sum()), self.T_STEPS))
            CTCFocusLoss._warned = True
        # (target offsets = exclusive prefix sums of the lengths, on the host with the rest of the encoding: on the device they
        # were a scan + two elementwise launches in front of every CTC kernel)
        return t.to(device), l.to(device), (torch.cumsum(l, 0) - l).to(torch.int32).to(device)

    def forward(self, sr_img, hr_img, label_strs=None, encoded=None):
        mse = K.mse_loss(sr_img, hr_img)
        rec = self.recognizer[0]
        if rec is None:
            return mse, mse, -1, -1
        if encoded is None:
            encoded = self.encode(label_strs, sr_img.device)
        gray = K.bicubic_gray(sr_img, 100)                  # parse_crnn_data, base.py:319-325
        logits = rec(gray)                                  # [26, B, 37]
        ctc = K.ctc_loss(logits, encoded[0], encoded[1], encoded[2] if len(encoded) > 2 else None)
        return mse + ctc, mse, -1, ctc

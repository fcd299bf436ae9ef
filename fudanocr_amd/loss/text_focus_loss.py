"""TextFocusLoss of the reference (scene-text-telescope/loss/text_focus_loss.py:41-104) on the HIP path:

    loss = mse(sr, hr) + 10 * L1(attention_map(hr), attention_map(sr)) + 0.0005 * weight_cross_entropy(pred(sr), gt)

with the frozen, eval-mode transformer recognizer (loss/transformer.py) applied to the luma of HR (no gradient) and of
SR (data gradient only, back into the SR network).  Returns the reference's 4-tuple (loss, mse_loss, attention_loss,
recognition_loss), `-1` sentinels when `args.text_focus` is off.

The reference loads ./dataset/mydata/pretrain_transformer.pth and ./dataset/mydata/confuse.pkl; neither ships with
it.  When the files are absent the recognizer gets the name-keyed deterministic weights (the same rule as every parity
fixture) and the confusion weight table is all ones (= plain cross-entropy): the maths, shapes and cost of the step are
the reference's, only the learned content is missing -- and that is logged."""
import logging
import os
import pickle

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ..sld import ops
from ..utils.util import str_filt as _str_filt
from ..utils.weight_fill import fill_module_
from .transformer import Transformer

standard_alphebet = "-0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def to_gray_tensor(tensor):
    """0.299 R + 0.587 G + 0.114 B (text_focus_loss.py:17-22): the luma kernel of parse_crnn_data at unchanged width"""
    return K.bicubic_gray(tensor, tensor.shape[3])


def str_filt(str_, voc_type):
    return _str_filt(str_, voc_type).lower()          # text_focus_loss.py:25-38 lower-cases once more at the end


def load_confuse_matrix(path="./dataset/mydata/confuse.pkl", allow_standin=None, used=None):
    """weight table of loss/weight_ce_loss.py:10-33 ([37, 37]; inverse confusion counts, lower/upper case merged).
    A missing file raises (the reference opens it at import, weight_ce_loss.py:35) unless stand-ins are allowed
    explicitly (stroke_focus_loss.standin_allowed)."""
    from .stroke_focus_loss import missing_asset, standin_allowed
    if not os.path.isfile(path):
        if not standin_allowed(allow_standin):
            raise missing_asset(path, "unit cross-entropy weights")
        logging.getLogger(__name__).warning("confuse.pkl not found (%s): weight_cross_entropy uses unit weights", path)
        if used is not None:
            used.append("confuse.pkl")
        return torch.ones(37, 37)
    data = pickle.load(open(path, "rb"))
    number, upper, lower = data[:10], data[10:36], data[36:]
    rearr = np.concatenate((np.ones((1, 62)), number, lower, upper), axis=0)
    rearr = np.concatenate((np.ones((63, 1)), rearr), axis=1)
    with np.errstate(divide="ignore"):
        rearr = 1 / rearr
    rearr[rearr == np.inf] = 1
    t = torch.Tensor(rearr)
    lower_alpha = "abcdefghijklmnopqrstuvwxyz"
    for i in range(63):
        for j in range(63):
            if i != j and standard_alphebet[j] in lower_alpha:
                t[i][j] = max(t[i][j], t[i][j + 26])
    return t[:37, :37].contiguous()


class TextFocusLoss(nn.Module):
    def __init__(self, args, transformer=None, weight_table=None, device="cuda", allow_standin=None):
        super().__init__()
        self.args = args
        self.allow_standin = allow_standin if allow_standin is not None else (
            True if getattr(args, "standin_assets", False) else None)
        self.standin_assets = []                          # names of the assets replaced by stand-ins (run config / log)
        self.english_alphabet = standard_alphebet
        self.english_dict = {c: i for i, c in enumerate(self.english_alphabet)}
        self.device = torch.device(device)
        self._transformer = [transformer]                 # not registered: stays out of state_dict / parameters()
        self._table = weight_table
        if getattr(args, "text_focus", False) and transformer is None:
            self.build_up_transformer()

    @property
    def transformer(self):
        return self._transformer[0]

    def build_up_transformer(self, path="./dataset/mydata/pretrain_transformer.pth"):
        t = Transformer()
        if os.path.isfile(path):
            sd = torch.load(path, map_location="cpu")
            t.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()})
        else:
            from .stroke_focus_loss import missing_asset, standin_allowed
            if not standin_allowed(self.allow_standin):
                raise missing_asset(path, "name-keyed deterministic recognizer weights")
            logging.getLogger(__name__).warning("pretrain_transformer.pth not found (%s): name-keyed weights", path)
            self.standin_assets.append(os.path.basename(path))
            fill_module_(t)
        t = t.to(self.device).eval()
        for p in t.parameters():
            p.requires_grad = False
        self._transformer[0] = t

    def weight_table(self):
        if self._table is None:
            self._table = load_confuse_matrix(allow_standin=self.allow_standin, used=self.standin_assets)
        if self._table.device != self.device:
            self._table = self._table.to(self.device).contiguous()
        return self._table

    def label_encoder(self, label):
        """text_focus_loss.py:62-81: teacher-forcing input shifted right by one, flat targets; CUDA tensors"""
        length = [len(i) for i in label]
        input_tensor = np.zeros((len(label), max(length)), dtype=np.int64)
        for i, s in enumerate(label):
            for j in range(length[i] - 1):
                input_tensor[i][j + 1] = self.english_dict[s[j]]
        text_gt = torch.tensor([self.english_dict[c] for s in label for c in s], dtype=torch.long)
        length_tensor = torch.tensor(length, dtype=torch.long).to(self.device)
        length_tensor._focr_host = length
        return length_tensor, torch.from_numpy(input_tensor).to(self.device), text_gt.to(self.device)

    # ---- recordable step (engine.TrainStep): the labels as a PaddedLabels batch, the forward as kernel launches only
    REPLAY_SAFE = True                 # forward(sr, hr, None, encoded) launches kernels only
    # label capacity of a recording: the batch's longest label rounded up to a multiple of this.  A padded position costs the
    # decoder 0.16 ms per 128 samples (profiles/r06_label_bucket_sweep.txt: B = 128 38.3 ms at 4, 38.9 at 8 / 16, 41.1 at 32);
    # a finer bucket costs recordings (engine.TrainStep.MAX_RECORDINGS, then eager steps)
    LABEL_BUCKET = int(os.environ.get("FOCR_LABEL_BUCKET", "4"))

    def encode(self, label, device=None, bucket=0):
        """the reference's filtering + `-` terminator (text_focus_loss.py:88) and label_encoder (:62-81) as one padded
        device tensor (loss/padded_labels.py); bucket = 0: capacity = the longest label, i.e. the reference's shapes"""
        from .padded_labels import PaddedLabels
        label = [str_filt(i, "lower") + "-" for i in label]
        return PaddedLabels.build([[self.english_dict[c] for c in s] for s in label], device or self.device, bucket)

    def encode_for_replay(self, label, device=None):
        if not getattr(self.args, "text_focus", False):
            return None
        return self.encode(label, device, self.LABEL_BUCKET)

    # ---- the HR branch (no gradient; depends on the HR images and the labels only) starts BEFORE the SR network's forward
    # pass, on the engine's side stream, which is idle until the backward pass (engine.TrainStep calls prefetch_hr in front
    # of the model; FOCR_HR_SIDE=0 switches it off).  B = 16: 9.27 -> 8.48 ms, B = 128: 36.97 -> 36.61 ms
    # (profiles/r06_hr_side_ab.txt).  The first call of a criterion runs the branch in line: the recognizer's lazily
    # prepared tables (fragment-ordered / BatchNorm-folded weights) are then made on the stream every later use is ordered
    # behind.
    def prefetch_hr(self, hr_img, encoded, side):
        if not getattr(self.args, "text_focus", False) or encoded is None or not getattr(self, "_hr_warm", False):
            return
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            _, amap = self.transformer.forward_padded(to_gray_tensor(hr_img), encoded.text_input)
            ev = torch.cuda.Event()
            ev.record(side)
        amap.record_stream(cur)
        self._hr_ready = (amap, ev, encoded)

    def forward(self, sr_img, hr_img, label, encoded=None):
        mse_loss = K.mse_loss(sr_img, hr_img)
        if not getattr(self.args, "text_focus", False):
            return mse_loss, mse_loss, -1, -1
        enc = encoded if encoded is not None else self.encode(label, sr_img.device)
        tr = self.transformer
        ready = self.__dict__.pop("_hr_ready", None)
        if ready is not None and ready[2] is enc:
            word_attention_map_gt = ready[0]
            torch.cuda.current_stream().wait_event(ready[1])
        else:
            with torch.no_grad():
                _, word_attention_map_gt = tr.forward_padded(to_gray_tensor(hr_img), enc.text_input)
            self._hr_warm = True
        sr_logits, word_attention_map_pred = tr.forward_padded(to_gray_tensor(sr_img), enc.text_input)
        # nn.L1Loss over the [B, 16, max(len), 256] maps and weight_cross_entropy over the sum(len) real positions
        # (text_focus_loss.py:92-93), read out of the padded layout
        attention_loss = ops.l1_loss_masked(word_attention_map_gt, word_attention_map_pred, enc.plan)
        recognition_loss = ops.weight_cross_entropy_masked(sr_logits.reshape(enc.batch * enc.cap, -1), enc.text_gt,
                                                           self.weight_table(), enc.plan)
        loss = mse_loss + attention_loss * 10 + recognition_loss * 0.0005
        return loss, mse_loss, attention_loss, recognition_loss

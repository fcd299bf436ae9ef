"""StrokeFocusLoss of text-gestalt (reference text-gestalt/loss/stroke_focus_loss.py:20-118) on the HIP path:

    loss = mse(sr, hr) + stroke_lambda * L1(attention_map(hr), attention_map(sr))

Labels are decomposed into stroke sequences (`english_decomposition.txt`: one "<character> <digits>" line per
character, strokes 1-9; `0` closes the word) and fed with teacher forcing to the frozen, eval-mode stroke-level
recognizer (loss/transformer_english_decomposition.py) on the luma of HR (no gradient) and of SR (data gradient back
into the SR network).  Returns the reference's 4-tuple (loss, mse_loss, attention_loss, recognition_loss = -1), with
`-1` sentinels for both terms when `args.text_focus` is off.

Neither ./dataset/mydata/english_decomposition.txt nor pretrain_transformer_stroke_decomposition.pth ships with the
reference.  As in the reference (stroke_focus_loss.py:31,45) a missing file raises FileNotFoundError: training the SR
network towards the attention maps of an untrained recognizer over a made-up stroke table is not a run anybody wants
by accident.  Benchmarks and tests opt in explicitly (`allow_standin=True`, `--standin_assets`, or
FOCR_ALLOW_STANDIN_ASSETS=1): the decomposition is then a seeded stand-in table (`standin_decomposition`: same format,
every alphanumeric character -> 1-4 strokes) and the recognizer gets the name-keyed deterministic weights -- the maths,
shapes and cost of the step are the reference's, only the learned content is missing; `standin_assets` lists what was
substituted and the harness writes it into the run's log / config."""
import logging
import os

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ..sld import ops
from ..utils.weight_fill import fill_module_
from .text_focus_loss import to_gray_tensor
from .transformer_english_decomposition import Transformer


def standin_decomposition(seed=2021):
    """character -> stroke digits (1-9), for 0-9a-zA-Z: the FORMAT of english_decomposition.txt with seeded content"""
    chars = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"
    rs = np.random.RandomState(seed)
    return {c: "".join(str(int(d)) for d in rs.randint(1, 10, size=int(rs.randint(1, 5)))) for c in chars}


def standin_allowed(flag=None):
    """explicit opt-in for stand-in assets: constructor argument, else FOCR_ALLOW_STANDIN_ASSETS=1"""
    return bool(flag) if flag is not None else os.environ.get("FOCR_ALLOW_STANDIN_ASSETS", "0") == "1"


def missing_asset(path, what):
    return FileNotFoundError("%s not found (%s). The reference needs this file too; pass --standin_assets (or "
                             "allow_standin=True / FOCR_ALLOW_STANDIN_ASSETS=1) to run on %s instead" % (
                                 os.path.basename(path), path, what))


def load_decomposition(path="./dataset/mydata/english_decomposition.txt", allow_standin=None, used=None):
    if not os.path.isfile(path):
        if not standin_allowed(allow_standin):
            raise missing_asset(path, "a seeded stand-in stroke table")
        logging.getLogger(__name__).warning("english_decomposition.txt not found (%s): seeded stand-in stroke table", path)
        if used is not None:
            used.append("english_decomposition.txt")
        return standin_decomposition()
    dic = {}
    for line in open(path, "r").readlines():
        character, sequence = line.strip().split()
        dic[character] = sequence
    return dic


class StrokeFocusLoss(nn.Module):
    correct_flag = False               # reference :99: the "select correct" branch is switched off

    def __init__(self, args, transformer=None, decomposition=None, device="cuda", allow_standin=None):
        super().__init__()
        self.args = args
        self.allow_standin = allow_standin if allow_standin is not None else (
            True if getattr(args, "standin_assets", False) else None)
        self.standin_assets = []                          # names of the assets replaced by stand-ins (run config / log)
        self.english_stroke_alphabet = "0123456789"
        self.english_stroke_dict = {c: i for i, c in enumerate(self.english_stroke_alphabet)}
        self.dic = dict(decomposition) if decomposition is not None else load_decomposition(
            allow_standin=self.allow_standin, used=self.standin_assets)
        self.device = torch.device(device)
        self._transformer = [transformer]                 # not registered: stays out of state_dict / parameters()
        if getattr(args, "text_focus", False) and transformer is None:
            self.build_up_transformer()

    @property
    def transformer(self):
        return self._transformer[0]

    def build_up_transformer(self, path="./dataset/mydata/pretrain_transformer_stroke_decomposition.pth"):
        t = Transformer()
        if os.path.isfile(path):
            sd = torch.load(path, map_location="cpu")
            t.load_state_dict({k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()})
        else:
            if not standin_allowed(self.allow_standin):
                raise missing_asset(path, "name-keyed deterministic recognizer weights")
            logging.getLogger(__name__).warning("pretrain_transformer_stroke_decomposition.pth not found (%s): "
                                                "name-keyed weights", path)
            self.standin_assets.append(os.path.basename(path))
            fill_module_(t)
        t = t.to(self.device).eval()
        for p in t.parameters():
            p.requires_grad = False
        self._transformer[0] = t

    def label_stroke_encoder(self, label):
        """reference :49-80: characters without a decomposition are skipped, `0` closes the word; teacher-forcing input
        shifted right by one, flat targets; CUDA tensors"""
        label = ["".join(self.dic[c] for c in one if c in self.dic) + "0" for one in label]
        length = [len(i) for i in label]
        input_tensor = np.zeros((len(label), max(length)), dtype=np.int64)
        for i, s in enumerate(label):
            for j in range(length[i] - 1):
                input_tensor[i][j + 1] = self.english_stroke_dict[s[j]]
        text_gt = torch.tensor([self.english_stroke_dict[c] for s in label for c in s], dtype=torch.long)
        length_tensor = torch.tensor(length, dtype=torch.long).to(self.device)
        length_tensor._focr_host = length
        return length_tensor, torch.from_numpy(input_tensor).to(self.device), text_gt.to(self.device)

    # ---- recordable step (engine.TrainStep): the labels as a PaddedLabels batch, the forward as kernel launches only
    REPLAY_SAFE = True                 # forward(sr, hr, None, encoded) launches kernels only
    # stroke sequences are long (a ten-letter word: ~30 strokes) but a padded position is as dear as in the text-focus loss
    # (B = 128: 41.3 ms at 4, 41.6 at 8, 42.8 at 16, 45.2 at 32: profiles/r06_label_bucket_sweep.txt)
    LABEL_BUCKET = int(os.environ.get("FOCR_LABEL_BUCKET", "8"))

    def encode(self, label, device=None, bucket=0):
        """label_stroke_encoder (reference :49-80) as one padded device tensor (loss/padded_labels.py)"""
        from .padded_labels import PaddedLabels
        label = ["".join(self.dic[c] for c in one if c in self.dic) + "0" for one in label]
        return PaddedLabels.build([[self.english_stroke_dict[c] for c in s] for s in label], device or self.device, bucket)

    def encode_for_replay(self, label, device=None):
        if not getattr(self.args, "text_focus", False):
            return None
        return self.encode(label, device, self.LABEL_BUCKET)

    def prefetch_hr(self, hr_img, encoded, side):
        """the HR branch on the engine's side stream, started before the SR network's forward pass (see
        text_focus_loss.TextFocusLoss.prefetch_hr)"""
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
        _, word_attention_map_pred = tr.forward_padded(to_gray_tensor(sr_img), enc.text_input)
        # nn.L1Loss over the [B, 16, max(len), 256] maps (reference :105), read out of the padded layout
        attention_loss = ops.l1_loss_masked(word_attention_map_gt, word_attention_map_pred, enc.plan)
        loss = mse_loss + attention_loss * float(getattr(self.args, "stroke_lambda", 50))
        return loss, mse_loss, attention_loss, -1

"""Label batches of the focus losses in a layout whose SHAPE does not depend on the labels.

The reference's label encoders (scene-text-telescope/loss/text_focus_loss.py:62-81, text-gestalt/loss/stroke_focus_loss.py:49-80)
build a [B, max(len)] teacher-forcing matrix and a flat [sum(len)] target vector: every tensor downstream changes shape from
batch to batch, and a step whose shapes change cannot be recorded once and re-issued (engine.TrainStep / csrc/replay.hip) --
the text-focus step then stays bound by the host at the reference's own batch size (README: --batch_size=16).

Here a batch is ONE int64 device tensor  [ B x L input | B x L targets, -1 = padding | plan ]  with L a capacity (the longest
label rounded up to a bucket) and plan = (longest label, number of real positions): the recognizer decodes all B x L
positions (a padded position only sees earlier ones through the causal mask and feeds nothing back), and the losses read the
real extents from `plan` on the device (csrc/sld_ops.hip focr_l1_masked_*, focr_weight_cross_entropy_masked_fwd).  One
host-to-device copy per step; same loss values as the reference layout up to summation order."""
import numpy as np
import torch


class PaddedLabels:
    PLAN = 4                               # int64 words behind the two matrices (2 used)

    def __init__(self, buf, batch, cap):
        n = batch * cap
        assert buf.numel() == 2 * n + self.PLAN and buf.dtype == torch.int64
        self.buf, self.batch, self.cap = buf, batch, cap
        self.text_input = buf[:n].view(batch, cap)
        self.text_gt = buf[n:2 * n]
        self.plan = buf[2 * n:]

    @classmethod
    def build(cls, ids, device, bucket=0):
        """ids: per sample the class ids of its label INCLUDING the closing symbol (len >= 1).  Teacher forcing: position
        j + 1 reads symbol j (position 0 reads class 0), the target of position j is symbol j."""
        batch = len(ids)
        lmax = max(len(s) for s in ids)
        cap = lmax if bucket <= 0 else (lmax + bucket - 1) // bucket * bucket
        arr = np.zeros(2 * batch * cap + cls.PLAN, dtype=np.int64)
        inp = arr[:batch * cap].reshape(batch, cap)
        gt = arr[batch * cap:2 * batch * cap].reshape(batch, cap)
        gt[:] = -1
        total = 0
        for i, s in enumerate(ids):
            n = len(s)
            inp[i, 1:n] = s[:n - 1]
            gt[i, :n] = s
            total += n
        arr[2 * batch * cap] = lmax
        arr[2 * batch * cap + 1] = total
        host = torch.from_numpy(arr)
        dev = torch.device(device)
        if dev.type == "cuda":
            host = host.pin_memory()
        out = cls(host.to(dev, non_blocking=True), batch, cap)
        out.lmax, out.total = lmax, total
        return out

    # ---- what engine.TrainStep needs to keep a recording's static copy of the labels
    def key(self):
        return ("padded", self.batch, self.cap)

    def static_like(self):
        return PaddedLabels(torch.zeros_like(self.buf), self.batch, self.cap)

    def fits(self, static):
        return static.batch == self.batch and static.cap == self.cap

    def copy_into(self, static):
        static.buf.copy_(self.buf, non_blocking=True)

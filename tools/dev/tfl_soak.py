"""TBSRN + TextFocusLoss as a long run with labels that change every step (lengths 1 .. 22: several capacity buckets, more
than the engine keeps recordings for): every step finite, recorded and eager steps interleave, device memory stops growing once
the recordings exist."""
import os, random, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fudanocr_amd import _lib
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
from fudanocr_amd.loss.transformer import Transformer
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
from fudanocr_amd.utils.weight_fill import fill_module_
_lib.load(); _lib.set_precision(3)
dev = torch.device("cuda", 0)
B, STEPS = int(os.environ.get("B", "16")), int(os.environ.get("STEPS", "400"))
net, _, _ = build_models(dev, "tbsrn", with_crnn=False)
tr = fill_module_(Transformer()).to(dev).eval()
for p in tr.parameters():
    p.requires_grad = False
crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, device=dev,
                     weight_table=torch.rand(37, 37, generator=torch.Generator().manual_seed(3)) + 0.5)
step = TrainStep(net, crit, dropout=True)
rng = random.Random(5)
alphabet = "0123456789abcdefghijklmnopqrstuvwxyz"
lr, hr, _ = make_batch(B, 99)
lr, hr = lr.to(dev), hr.to(dev)
mem, rec_steps, losses = [], 0, []
for s in range(STEPS):
    top = rng.choice((3, 6, 9, 12, 15, 18, 22))
    labels = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, top))) for _ in range(B)]
    step.recorded = None
    out = step(lr, hr, labels)
    rec_steps += int(step.recorded is not None)
    if s % 50 == 49 or s == STEPS - 1:
        torch.cuda.synchronize()
        losses.append(float(out["loss"]))
        mem.append(torch.cuda.memory_reserved() / 2 ** 30)
        print("step %4d  loss %.5f  recorded steps so far %d  recordings %d  reserved %.2f GiB" % (s + 1, losses[-1], rec_steps, len(step._recs), mem[-1]), flush=True)
assert all(l == l and abs(l) < 1e3 for l in losses), losses
assert len(step._recs) <= step.MAX_RECORDINGS
assert mem[-1] <= mem[len(mem) // 2] + 0.25, mem          # no growth over the second half of the run
print("ok: %d of %d steps re-issued from %d recordings" % (rec_steps, STEPS, len(step._recs)))

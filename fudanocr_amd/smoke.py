"""One tiny training step of the flagship path (TBSRN + frozen CRNN + CTC) on cuda:0, checked
against the CPU oracle (fp32 torch restatement pinned to the reference's golden vectors)."""
import torch


def build_models(device, arch="tbsrn", with_crnn=True, mask=False):
    from .loss.ctc_focus_loss import CTCFocusLoss
    from .model import tbsrn
    from .model.crnn import crnn
    from .utils.weight_fill import fill_module_
    if arch == "tbsrn":
        net = tbsrn.TBSRN(STN=True, mask=mask)
    else:
        from .model import tsrn
        net = tsrn.TSRN(STN=True, mask=mask)
    fill_module_(net)
    net = net.to(device)
    rec = None
    if with_crnn:
        rec = crnn.CRNN(32, 1, 37, 256)
        fill_module_(rec)
        rec = rec.to(device).eval()
        for p in rec.parameters():
            p.requires_grad = False
    return net, rec, CTCFocusLoss(rec)


def set_dropout(net, on):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.train(on)


def run(batch=4):
    from . import _lib
    from .engine import TrainStep
    from .utils.synth import make_batch
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU")
    _lib.load()
    dev = torch.device("cuda:0")
    net, rec, crit = build_models(dev)
    lr, hr, labels = make_batch(batch, 1234)
    step = TrainStep(net, crit, dropout=False)
    out = step(lr.to(dev), hr.to(dev), labels)
    torch.cuda.synchronize()
    # ---- oracle on the same inputs / weights ----
    from oracle import sr_oracle as O
    from .utils.weight_fill import fill_dict_
    P = O.make_params(O.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    C = O.make_params(O.schema_crnn(), requires_grad=False)
    fill_dict_(C)
    opt = O.AdamState([v for v in P.values() if v.requires_grad])
    tgt, tlen = O.encode_labels(labels)
    ref = O.train_step(P, opt, "tbsrn", lr, hr, C, tgt, tlen)
    sr_err = (out["sr"].cpu() - ref["sr"]).abs().max().item() / ref["sr"].abs().max().item()
    l_err = abs(out["loss"].item() - ref["loss"]) / abs(ref["loss"])
    g_err = abs(step.opt.grad_norm().item() - ref["grad_norm"]) / ref["grad_norm"]
    print("smoke: loss %.6f (oracle %.6f, rel %.2e)  sr rel-to-max err %.2e  grad-norm rel err %.2e"
          % (out["loss"].item(), ref["loss"], l_err, sr_err, g_err))
    assert sr_err < 1e-3 and l_err < 1e-3 and g_err < 2e-2, "smoke parity failed"
    return True

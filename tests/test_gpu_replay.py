"""GPU tests of the recorded step (csrc/replay.hip, fudanocr_amd/replay.py, engine.TrainStep replay mode): the launch list the
library builds from a captured graph computes what the eager launches compute -- on a synthetic multi-stream graph and on
the whole training step (reference step: interfaces/super_resolution.py:79-84), bit for bit, dropout included."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fudanocr_amd.utils.synth import make_batch          # noqa: E402


def build(arch="tbsrn", with_crnn=True):
    from fudanocr_amd.smoke import build_models
    return build_models(torch.device("cuda:0"), arch, with_crnn=with_crnn)


def test_replay_multi_stream_graph():
    """fork / join over three streams with kernel and memset nodes: every replay reproduces the eager result on fresh
    inputs, and the launch list keeps the captured concurrency (more than one lane); a graph with a node the library cannot
    re-issue faithfully (hipMemcpyAsync's 1-D copy node) is REFUSED, not guessed at"""
    from fudanocr_amd import _lib, replay
    _lib.load()
    dev = torch.device("cuda", 0)
    x = torch.zeros(1 << 20, device=dev)
    a, b, c, d, out = (torch.zeros_like(x) for _ in range(5))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def body():
        cur = torch.cuda.current_stream()
        a.zero_()                                  # memset node
        a.add_(x)
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            torch.add(a, 0.0, out=b)               # (a copy KERNEL: hipMemcpyAsync's 1-D graph node cannot be read back)
            for _ in range(3):
                b.mul_(1.5).add_(1.0)
        with torch.cuda.stream(s2):
            torch.add(a, 0.0, out=c)
            c.mul_(a).sub_(2.0)
        torch.mul(a, 2.0, out=d)                   # stays on the origin stream, concurrent with both branches' tails
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        torch.add(b, c, out=out)
        out.add_(d)
        return out

    def expect(v):
        a_ = v.clone()
        b_ = a_.clone()
        for _ in range(3):
            b_ = b_ * 1.5 + 1.0
        c_ = a_ * a_ - 2.0
        return b_ + c_ + a_ * 2.0

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                  # warm-up outside the recording
        body()
    torch.cuda.synchronize()
    rec, res = replay.record(body)
    assert rec.info["kernels"] >= 10 and rec.info["lanes"] >= 2 and rec.info["waits"] >= 2, rec.info
    assert rec.info["nodes"] == len(rec.node_names()) == len(rec.node_lanes())
    for i in range(4):
        x.copy_(torch.randn(1 << 20, generator=torch.Generator().manual_seed(i)).to(dev))
        rec.launch()
        torch.cuda.synchronize()
        assert torch.equal(res, expect(x)), i
    # timing probes: one event pair around every node that matches
    n = rec.probe("elementwise", depth=3)
    assert n >= 5
    for _ in range(5):
        rec.launch()
    times = rec.probe_read()
    assert len(times) == n and all(ms > 0 and cnt == 3 for _, ms, cnt in times)
    rec.close()

    def with_copy():
        b.copy_(a)
        return b

    with torch.cuda.stream(side):
        with_copy()
    torch.cuda.synchronize()
    try:
        rec2, _ = replay.record(with_copy)
    except RuntimeError as e:
        assert "memcpy" in str(e)
    else:                                          # a runtime that describes the node fully: then it must replay correctly
        a.fill_(3.0)
        rec2.launch()
        torch.cuda.synchronize()
        assert torch.equal(b, a)


def _snapshot(step, net):
    bufs = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
    return (step.flat.flat_param.clone(), step.opt.m.clone(), step.opt.v.clone(), step.state.buf.clone(), step.opt.t, bufs)


def _restore(step, net, snap):
    p, m, v, st, t, bufs = snap
    step.flat.flat_param.copy_(p)
    step.opt.m.copy_(m)
    step.opt.v.copy_(v)
    step.state.buf.copy_(st)
    step.opt.t = t
    sd = net.state_dict()
    for k, b in bufs.items():
        sd[k].copy_(b)


@pytest.mark.parametrize("arch,with_crnn", [("tbsrn", True), ("tbsrn", False), ("tsrn", True)], ids=["c3", "c2", "c1"])
def test_recorded_step_equals_eager(arch, with_crnn):
    """Steps 3 .. 7 of a run on a cycle of three batches, dropout ON, each taken TWICE from the same state (parameters, Adam
    moments, step state, BatchNorm running statistics restored in between): once re-issued from the library's recording,
    once launched from Python.  Same loss (same dropout bits: seed and device epoch), same pre-clip gradient norm, same
    update, same SR output, same running statistics -- up to the last-bit spread two launches of the same step have
    (fp32 atomics in a few small backward kernels; Adam turns a sign flip of a noise-level gradient into a 2 lr step of
    that element, hence the displacement bound).  Comparing whole trajectories instead is useless: two EAGER runs drift
    apart by 1e-3 within five steps at this batch size.  Covers the device-resident step state (dropout epoch, Adam step
    count + bias corrections), the static input / label buffers and the LSTM scan's recorded workspace flags."""
    from fudanocr_amd.engine import TrainStep
    lr_ = 1e-4
    net, rec, crit = build(arch, with_crnn)
    step = TrainStep(net, crit, lr=lr_, dropout=True, replay=True, seed=77)
    seen_losses = []
    for s in range(7):
        l, h, labels = make_batch(4, 1234 + s % 3)
        l, h = l.cuda(), h.cuda()
        labels = labels if with_crnn else None
        if s < 2:
            seen_losses.append(step(l, h, labels)["loss"].item())          # the engine's own eager warm-up steps
            continue
        snap = _snapshot(step, net)
        res = {}
        for how in ("replay", "eager"):
            _restore(step, net, snap)
            step.replay = how == "replay"
            out = step(l, h, labels)
            res[how] = (out["loss"].item(), step.opt.grad_norm().item(), step.flat.flat_param.clone(), out["sr"].clone(),
                        torch.cat([v.double().flatten() for k, v in net.state_dict().items() if "running_" in k]),
                        step.state.counts(), step.opt.t)
        step.replay = True
        assert step.recorded is not None and step.recorded.info["kernels"] > 100 and step.recorded.info["lanes"] >= 2
        (l_r, g_r, p_r, sr_r, bn_r, cnt_r, t_r), (l_e, g_e, p_e, sr_e, bn_e, cnt_e, t_e) = res["replay"], res["eager"]
        assert cnt_r == cnt_e == (s + 1, s + 1) and t_r == t_e == s + 1
        assert abs(l_r - l_e) <= 2e-6 * abs(l_e), (s, l_r, l_e)
        assert abs(g_r - g_e) <= 1e-4 * abs(g_e), (s, g_r, g_e)
        d = (p_r - p_e).abs()
        moved = (p_e - snap[0]).abs()
        assert float(d.max()) <= 2.5 * lr_ and float(d.mean()) <= 0.02 * float(moved.mean()), (s, d.max(), d.mean(), moved.mean())
        assert float((sr_r - sr_e).abs().max()) <= 1e-5 * float(sr_e.abs().max()), s
        assert float((bn_r - bn_e).abs().max()) <= 1e-5 * float(bn_e.abs().max()), s
        seen_losses.append(l_e)
    # dropout really was on and really changes from step to step: the same batch (steps 0, 3, 6) never repeats a loss
    assert len({seen_losses[0], seen_losses[3], seen_losses[6]}) == 3


def test_recorded_step_fresh_dropout_every_replay():
    """learning rate 0: parameters never move, the batch is fixed, so the only thing that can change the loss from one
    replay to the next is the dropout epoch on the device"""
    from fudanocr_amd.engine import TrainStep
    net, rec, crit = build()
    step = TrainStep(net, crit, lr=0.0, dropout=True, replay=True, seed=5)
    l, h, labels = make_batch(4, 1234)
    l, h = l.cuda(), h.cuda()
    enc = crit.encode(labels, l.device)
    losses = [step(l, h, encoded=enc)["loss"].item() for _ in range(6)]
    assert step.recorded is not None
    assert len(set(losses)) == 6, losses
    e, t = step.state.counts()
    assert (e, t) == (6, 6) and step.opt.t == 6
    # the static inputs can be written in place of being copied into
    st = step.recorded_inputs(l, h, enc)
    assert st is not None and st[0].shape == l.shape and st[2][1].shape == enc[1].shape
    out = step(st[0], st[1], encoded=st[2])
    assert out["loss"].item() not in losses


def test_recorded_step_shape_change_and_eval_between():
    """a batch of another size steps eagerly (then gets its own recording), eval in between sees the trained statistics
    (the BatchNorm eval cache is dropped after every replay), and switching replay off mid-run keeps stepping"""
    from fudanocr_amd.engine import TrainStep
    net, rec, crit = build()
    step = TrainStep(net, crit, dropout=False, replay=True)
    probe = make_batch(4, 99)[0].cuda()
    for s in range(4):
        l, h, labels = make_batch(4, 10 + s)
        step(l.cuda(), h.cuda(), labels)
    assert step.recorded is not None and len(step._recs) == 1
    net.eval()
    with torch.no_grad():
        e1 = net(probe).clone()
    l, h, labels = make_batch(8, 20)
    step(l.cuda(), h.cuda(), labels)                   # other shape: eager warm-up of a second signature
    assert len(step._recs) == 1
    for s in range(2):
        l, h, labels = make_batch(4, 30 + s)
        step(l.cuda(), h.cuda(), labels)
    net.eval()
    with torch.no_grad():
        e2 = net(probe).clone()
    assert not torch.equal(e1, e2)
    fresh, _, _ = build()
    fresh.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    fresh.eval()
    with torch.no_grad():
        e3 = fresh(probe)
    assert torch.equal(e2, e3), (e2 - e3).abs().max().item()
    # a changed learning rate is part of the recording's signature: the next steps warm up eagerly and record again -- the old
    # recording (which carries the old rate as a kernel argument) is not reused
    def three_steps(seed0):
        before = step.flat.flat_param.clone()
        for s in range(3):
            l_, h_, lab_ = make_batch(4, seed0 + s)
            step(l_.cuda(), h_.cuda(), lab_)
        return float((step.flat.flat_param - before).abs().mean())

    full = three_steps(50)                      # three replayed steps at the recorded rate 1e-4
    assert len(step._recs) == 1
    step.opt.lr = 2.5e-5
    quarter = three_steps(60)                   # two eager warm-ups + a new recording at a quarter of the rate
    assert len(step._recs) == 2
    assert 0.15 * full < quarter < 0.4 * full, (full, quarter)
    step.replay = False
    l, h, labels = make_batch(4, 40)
    out = step(l.cuda(), h.cuda(), labels)
    assert torch.isfinite(out["loss"]).item()


def test_mode3_trains_like_mode1():
    """bench.py's default arithmetic (precision mode 3: single-bf16 data-gradient products) against the fp32-equivalent one
    (mode 1: split products at every site) as TRAINING runs: 300 steps of the c3 step at B = 128 on a fixed 8-batch cycle,
    dropout off, identical initial weights, two runs per mode (tools/mode3_equivalence.py; committed artefact with three
    runs per mode: profiles/r06_mode3_equivalence.json).  The loss curves agree to 1e-3 at EVERY step, and everything else
    -- parameter displacement direction, PSNR of the eval forward -- differs between the modes no more than it differs
    between two runs of the SAME mode (two identical runs drift apart: fp32 atomics + Adam's normalised steps)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import mode3_equivalence as M
    res = M.main(steps=300, batch=128, repeats=2)
    assert res["replayed"]
    x, w1, w3 = res["mode1_vs_mode3"], res["within_mode1"], res["within_mode3"]
    assert res["loss_first_last"]["mode1"][1] < 0.97 * res["loss_first_last"]["mode1"][0]       # it does train
    assert x["max_rel_loss_diff"] <= 1e-3, x
    within_cos = min(w1["mean_displacement_cosine"], w3["mean_displacement_cosine"])
    assert x["mean_displacement_cosine"] >= within_cos - 0.01 and x["mean_displacement_cosine"] >= 0.98, (x, w1, w3)
    assert x["worst_displacement_cosine"] >= min(w1["worst_displacement_cosine"], w3["worst_displacement_cosine"]) - 0.03
    assert x["psnr_diff_db_mean"] <= 2 * max(w1["psnr_diff_db_max"], w3["psnr_diff_db_max"]) + 0.02, (x, w1, w3)


def test_sld_recorded_step_equals_eager():
    """the stroke-level-decomposition recognizer's step (reference stroke-level-decomposition/train.py:63-77) recorded and
    re-issued from the library against the same step launched from Python, from the same restored state, dropout on:
    loss, predictions and the Adadelta update agree to the last-bit spread of two launches"""
    from fudanocr_amd.sld import util as sld_util
    from fudanocr_amd.sld.engine import SLDTrainStep
    from fudanocr_amd.sld.model.transformer import Transformer
    from fudanocr_amd.sld.synth import make_sld_batch
    from fudanocr_amd.utils.weight_fill import fill_module_
    dev = torch.device("cuda", 0)
    net = fill_module_(Transformer("stroke")).to(dev)
    step = SLDTrainStep(net, dropout=True, replay=True)
    image, labels = make_sld_batch(4, 77)
    image = image.to(dev)
    length, text_input, text_gt, _ = sld_util.converter("stroke", labels, device=dev, strokes=True)
    assert length._focr_idx.numel() == text_gt.numel()
    for _ in range(2):
        step(image, length, text_input, text_gt)
    bufs = lambda: {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}  # noqa: E731
    for s_ in range(3):
        snap = (step.flat.flat_param.clone(), step.opt.sq.clone(), step.opt.acc.clone(), step.state.buf.clone(), bufs())
        res = {}
        for how in ("replay", "eager"):
            step.flat.flat_param.copy_(snap[0]); step.opt.sq.copy_(snap[1]); step.opt.acc.copy_(snap[2])
            step.state.buf.copy_(snap[3])
            sd = net.state_dict()
            for k, b in snap[4].items():
                sd[k].copy_(b)
            step.replay = how == "replay"
            out = step(image, length, text_input, text_gt)
            res[how] = (out["loss"].item(), out["pred"].clone(), step.flat.flat_param.clone())
        step.replay = True
        assert step.recorded is not None and step.recorded.info["kernels"] > 100
        (l_r, p_r, w_r), (l_e, p_e, w_e) = res["replay"], res["eager"]
        assert abs(l_r - l_e) <= 5e-6 * abs(l_e), (s_, l_r, l_e)
        assert float((p_r - p_e).abs().max()) <= 1e-4 * float(p_e.abs().max()), s_
        moved = (w_e - snap[0]).abs()
        assert float((w_r - w_e).abs().mean()) <= 0.02 * float(moved.mean()), s_


@pytest.mark.parametrize("kind", ["tfl", "sfl"])
def test_recorded_focus_step_equals_eager(kind):
    """The reference's real training criteria (main.py --text_focus: TextFocusLoss; text-gestalt: StrokeFocusLoss) as a
    RECORDED step: labels padded to a capacity bucket (loss/padded_labels.py) so that no launch depends on the batch's
    labels.  A cycle of batches with DIFFERENT labels -- different lengths inside one bucket, and a second bucket -- each
    step taken twice from the same restored state, re-issued from the recording and launched from Python: same loss terms,
    gradient norm, update and SR output.  Dropout on."""
    import types
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.loss.stroke_focus_loss import StrokeFocusLoss, standin_decomposition
    from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
    from fudanocr_amd.utils.weight_fill import fill_module_
    dev = torch.device("cuda", 0)
    net, _, _ = build("tbsrn", False)
    if kind == "tfl":
        from fudanocr_amd.loss.transformer import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, device=dev,
                             weight_table=torch.rand(37, 37, generator=torch.Generator().manual_seed(3)) + 0.5)
        long_label = "abcdefghij0123"                        # 15 symbols with the terminator: the second bucket (16)
    else:
        from fudanocr_amd.loss.transformer_english_decomposition import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = StrokeFocusLoss(types.SimpleNamespace(text_focus=True, stroke_lambda=50), transformer=tr, device=dev,
                               decomposition=standin_decomposition())
        long_label = "abcdefghijklmnopqrstuvwxyz"            # well over 32 strokes: the second bucket
    for p in tr.parameters():
        p.requires_grad = False
    crit.LABEL_BUCKET = 8 if kind == "tfl" else 32           # (pinned: the schedule below is built around two capacities)
    lr_ = 1e-4
    step = TrainStep(net, crit, lr=lr_, dropout=True, replay=True, seed=5)
    caps = set()
    n_replayed = 0
    short = [["ab1", "c", "de", "f9z"], ["q", "rs7", "tu", "v"], ["w0", "xyz", "a", "bc"]]       # <= 4 symbols: first bucket
    for s in range(14):
        l, h, _ = make_batch(4, 500 + s % 4)
        labels = [long_label, "ab", "c", "d1"] if s % 3 == 2 else short[(s // 3 + s % 3) % 3]
        l, h = l.cuda(), h.cuda()
        caps.add(crit.encode_for_replay(labels, dev).cap)
        if s < 2:
            step(l, h, labels)           # the engine's lazy tables (attention keep-bit buffers, fragment / flip tables) fill here
            continue
        snap = _snapshot(step, net)
        res = {}
        for how in ("replay", "eager"):
            _restore(step, net, snap)
            step.replay = how == "replay"
            step.recorded = None
            out = step(l, h, labels)
            res[how] = (out["loss"].item(), out["mse"].item(), step.opt.grad_norm().item(), step.flat.flat_param.clone(),
                        out["sr"].clone(), step.recorded is not None)
        step.replay = True
        (l_r, m_r, g_r, p_r, sr_r, was_rec), (l_e, m_e, g_e, p_e, sr_e, _) = res["replay"], res["eager"]
        n_replayed += int(was_rec)
        assert abs(l_r - l_e) <= 2e-6 * abs(l_e) and abs(m_r - m_e) <= 2e-6 * abs(m_e), (s, l_r, l_e)
        assert abs(g_r - g_e) <= 1e-4 * abs(g_e), (s, g_r, g_e)
        d = (p_r - p_e).abs()
        moved = (p_e - snap[0]).abs()
        assert float(d.max()) <= 2.5 * lr_ and float(d.mean()) <= 0.02 * float(moved.mean()), (s, d.max(), d.mean())
        assert float((sr_r - sr_e).abs().max()) <= 1e-5 * float(sr_e.abs().max()), s
    assert len(caps) == 2, caps                                # two buckets were exercised ...
    assert n_replayed >= 4 and len(step._recs) == 2, (n_replayed, len(step._recs))     # ... and both were recorded and re-issued

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


def _run_steps(arch, with_crnn, steps, replay_on, batch=4, lr=1e-4):
    from fudanocr_amd.engine import TrainStep
    net, rec, crit = build(arch, with_crnn)
    step = TrainStep(net, crit, lr=lr, dropout=True, replay=replay_on, seed=77)
    rows = []
    for s in range(steps):
        l, h, labels = make_batch(batch, 1234 + s % 3)
        before = step.flat.flat_param.clone()
        out = step(l.cuda(), h.cuda(), labels if with_crnn else None)
        moved = (step.flat.flat_param - before).abs()
        rows.append((out["loss"].item(), step.opt.grad_norm().item(), moved.mean().item(), moved.max().item()))
    torch.cuda.synchronize()
    if replay_on:
        assert step.recorded is not None, "the engine never replayed"
        assert step.recorded.info["kernels"] > 100 and step.recorded.info["lanes"] >= 2, step.recorded.info
        assert step.state.counts() == (steps, steps)
    else:
        assert step.recorded is None
    return torch.tensor(rows, dtype=torch.float64)


@pytest.mark.parametrize("arch,with_crnn", [("tbsrn", True), ("tbsrn", False), ("tsrn", True)], ids=["c3", "c2", "c1"])
def test_recorded_step_equals_eager(arch, with_crnn):
    """Seven optimisation steps on a cycle of three batches, dropout ON, from identical weights and the same seed: the engine
    that records its third step and replays it from the library follows the engine that launches every step from Python as
    closely as a SECOND eager engine does.  (Two eager runs differ in the last bits -- fp32 atomics in a few small backward
    kernels --, and Adam's normalised steps plus the TPS warp on noise images amplify that from step to step: the
    run-to-run spread is measured here and is the yardstick.)  Per step: loss, pre-clip gradient norm, and the mean / max
    parameter displacement (a stale Adam step count would scale every update: 1 / (1 - 0.5^t) is 1.14 at t = 3 against 2
    at t = 1).  Covers the device-resident step state (dropout epoch, Adam step count + bias corrections), the static
    input / label buffers and the LSTM scan's recorded workspace flags."""
    a = _run_steps(arch, with_crnn, 7, False)
    b = _run_steps(arch, with_crnn, 7, False)
    c = _run_steps(arch, with_crnn, 7, True)
    rel = lambda u, v: ((u - v).abs() / u.abs().clamp_min(1e-30))          # noqa: E731
    noise, got = rel(a, b), rel(a, c)
    assert torch.equal(a[:2], c[:2]) or float(got[:2].max()) <= float(4 * noise[:2].max() + 1e-6)     # both eager there
    for s in range(7):
        for j, name in enumerate(("loss", "grad-norm")):
            # (one eager pair is a coarse sample of a spread that grows chaotically with the step index)
            assert float(got[s, j]) <= 10 * float(noise[:s + 1, j].max()) + 2e-5 * (s + 1), (s, name, got[s, j], noise[s, j])
        assert float(got[s, 2]) <= 0.01 + 4 * float(noise[s, 2]), (s, "mean displacement", a[s, 2], c[s, 2])
        assert float(got[s, 3]) <= 0.05 + 4 * float(noise[s, 3]), (s, "max displacement", a[s, 3], c[s, 3])
    # the first replayed step (index 2) starts from bit-identical state up to that noise: dropout masks of (seed, epoch 3)
    # are the same bits in both engines -- a different mask moves the loss by 1e-2
    assert float(got[2, 0]) < 1e-5, got[2]
    # dropout really was on and really changes from step to step: the same batch (steps 0, 3, 6) never repeats a loss
    assert len({float(a[0, 0]), float(a[3, 0]), float(a[6, 0])}) == 3


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
    step.replay = False
    l, h, labels = make_batch(4, 40)
    out = step(l.cuda(), h.cuda(), labels)
    assert torch.isfinite(out["loss"]).item()

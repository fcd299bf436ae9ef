"""Row N2 (BASELINE configs[4]): stroke-level-decomposition transformer recognizer.
CPU: the oracle restatement against fixture F7 (generated from the imported reference, tools/make_golden_sld.py).
GPU (-m gpu): the HIP product model against the fixture and the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from fudanocr_amd.sld.synth import make_sld_batch
from fudanocr_amd.utils.weight_fill import fill_dict_


def _rel(got, ref):
    got, ref = torch.as_tensor(np.asarray(got)).double(), torch.as_tensor(np.asarray(ref)).double()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def _fx(golden_dir):
    return (np.load(os.path.join(golden_dir, "sld_step.npz")), json.load(open(os.path.join(golden_dir, "sld_schema.json"))),
            json.load(open(os.path.join(golden_dir, "sld_traj2.json"))),
            json.load(open(os.path.join(golden_dir, "sld_gradnorms.json"))))


def test_sld_oracle_matches_reference_fixture(golden_dir):
    from oracle import sld_oracle as O
    g, sc, traj, gn = _fx(golden_dir)
    P = O.make_params()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in P.items()] == sc["schema"]
    fill_dict_({k: v.data for k, v in P.items()})
    image, _ = make_sld_batch(4, 1234)
    length, text_input, text_gt = O.converter_stroke(sc["labels"], sc["table"])
    assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
    opt = O.AdadeltaState([v for k, v in P.items() if v.requires_grad])
    out = O.train_step(P, opt, image, length, text_input, text_gt)
    assert _rel(out["pred"], g["pred"]) < 1e-4
    assert abs(out["loss"] - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert _rel(out["conv"][:, ::64, ::4, ::4], g["conv_sub"]) < 1e-4
    assert _rel(out["map"][:, :, :, ::16], g["map_sub"]) < 1e-4
    assert _rel(P["encoder.bn1.running_mean"], g["bn1_rm"]) < 1e-5 and _rel(P["encoder.bn1.running_var"], g["bn1_rv"]) < 1e-5
    assert abs(out["grad_norm"] - traj["grad_norm"][0]) < 1e-3 * traj["grad_norm"][0]
    out2 = O.train_step(P, opt, image, length, text_input, text_gt)
    assert abs(out2["loss"] - traj["loss"][1]) < 2e-2 * traj["loss"][1]       # Adadelta(lr=1) steps ~ sign(g)*3e-3
    dead = [k for k, v in gn.items() if v is None]
    assert sorted(dead) == sorted(k for k in P if "compress_attention_linear" in k)


# ------------------------------------------------------------------------------------------------ GPU
def _build_gpu():
    from fudanocr_amd.sld.model.transformer import Transformer
    from fudanocr_amd.utils.weight_fill import fill_module_
    m = Transformer("stroke")
    fill_module_(m)
    return m.cuda()


@pytest.mark.gpu
def test_sld_ops_against_float64():
    """the SLD-only kernels through their autograd wrappers vs float64 torch: attention (causal / cross, few queries,
    256-wide heads), any-width LayerNorm, relu(a+b), embedding, ragged gather, cross-entropy, Adadelta"""
    from fudanocr_amd import kernels as K
    from fudanocr_amd.sld import ops
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, generator=g, dtype=torch.float64) * 2 - 1)     # noqa: E731
    for lq, lk, causal in ((7, 7, True), (30, 256, False), (1, 1, True)):
        q, k, v = rnd(3, lq, 1024).requires_grad_(True), rnd(3, lk, 1024).requires_grad_(True), rnd(3, lk, 1024).requires_grad_(True)
        qh, kh, vh = (t.view(3, -1, 4, 256).transpose(1, 2) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / 16.0
        if causal:
            s = s.masked_fill(~torch.tril(torch.ones(lq, lk, dtype=torch.bool)), float("-inf"))
        p = torch.softmax(s, -1)
        o = (p @ vh).transpose(1, 2).reshape(3, lq, 1024)
        go = rnd(3, lq, 1024)
        o.backward(go)
        qd, kd, vd = (t.detach().float().cuda().requires_grad_(True) for t in (q, k, v))
        od, amap = ops.small_attention(qd, kd, vd, 4, causal=causal)
        assert (od.cpu().double() - o).abs().max() < 2e-5 and (amap.cpu().double() - p).abs().max() < 1e-5
        od.backward(go.float().cuda())
        for got, ref in ((qd.grad, q.grad), (kd.grad, k.grad), (vd.grad, v.grad)):
            assert (got.cpu().double() - ref).abs().max() < 2e-5 * (1 + ref.abs().max())
    # 16 heads x 64 (the text- / stroke-focus recognizers' decoder): one block per (batch, head) over all query rows
    # (csrc/sld_ops.hip small_attn_*_rows64_kernel), causal and cross forms, a gradient through the returned map as well
    for lq, lk, causal in ((11, 11, True), (11, 256, False), (3, 130, False), (1, 1, True)):
        q, k, v = rnd(2, lq, 1024).requires_grad_(True), rnd(2, lk, 1024).requires_grad_(True), rnd(2, lk, 1024).requires_grad_(True)
        qh, kh, vh = (t.view(2, -1, 16, 64).transpose(1, 2) for t in (q, k, v))
        s = qh @ kh.transpose(-1, -2) / 8.0
        if causal:
            s = s.masked_fill(~torch.tril(torch.ones(lq, lk, dtype=torch.bool)), float("-inf"))
        p = torch.softmax(s, -1)
        o = (p @ vh).transpose(1, 2).reshape(2, lq, 1024)
        go, gm = rnd(2, lq, 1024), rnd(2, 16, lq, lk) * 0.1
        ((o * go).sum() + (p * gm).sum()).backward()
        qd, kd, vd = (t.detach().float().cuda().requires_grad_(True) for t in (q, k, v))
        od, amap = ops.small_attention(qd, kd, vd, 16, causal=causal)
        assert (od.cpu().double() - o).abs().max() < 2e-5 and (amap.cpu().double() - p).abs().max() < 1e-5
        ((od * go.float().cuda()).sum() + (amap * gm.float().cuda()).sum()).backward()
        for got, ref in ((qd.grad, q.grad), (kd.grad, k.grad), (vd.grad, v.grad)):
            assert (got.cpu().double() - ref).abs().max() < 2e-5 * (1 + ref.abs().max())
    # dropout on the probabilities: expectation and backward consistency through the returned map
    qd, kd, vd = (rnd(2, 5, 1024).float().cuda().requires_grad_(True) for _ in range(3))
    od, amap = ops.small_attention(qd, kd, vd, 4, causal=False, p_drop=0.5)
    keep = (amap > 0).float().mean().item()
    assert 0.3 < keep < 0.7
    ref = (amap @ vd.detach().view(2, 5, 4, 256).transpose(1, 2)).transpose(1, 2).reshape(2, 5, 1024)
    assert (od - ref).abs().max() < 1e-5
    # LayerNorm, D = 1024
    x, r = rnd(9, 1024).requires_grad_(True), rnd(9, 1024).requires_grad_(True)
    a, b = (rnd(1024) + 1.5).requires_grad_(True), rnd(1024).requires_grad_(True)
    z = x + r
    y = a * (z - z.mean(-1, keepdim=True)) / (z.std(-1, keepdim=True) + 1e-6) + b
    gy = rnd(9, 1024)
    y.backward(gy)
    xd, rd, ad, bd = (t.detach().float().cuda().requires_grad_(True) for t in (x, r, a, b))
    yd = K.layernorm_std(xd, ad, bd, residual=rd)
    yd.backward(gy.float().cuda())
    assert (yd.cpu().double() - y).abs().max() < 2e-5
    for got, ref in ((xd.grad, x.grad), (rd.grad, r.grad), (ad.grad, a.grad), (bd.grad, b.grad)):
        assert (got.cpu().double() - ref).abs().max() < 5e-5 * (1 + ref.abs().max())
    # relu(a + b), embedding, gather, cross-entropy
    a2, b2 = rnd(4, 8, 8, 64).requires_grad_(True), rnd(4, 8, 8, 64).requires_grad_(True)
    torch.relu(a2 + b2).backward(gy2 := rnd(4, 8, 8, 64))
    ad2, bd2 = a2.detach().float().cuda().requires_grad_(True), b2.detach().float().cuda().requires_grad_(True)
    yr = ops.add_relu(ad2, bd2)
    yr.backward(gy2.float().cuda())
    assert (yr.cpu().double() - torch.relu(a2 + b2)).abs().max() < 1e-6 and (ad2.grad.cpu().double() - a2.grad).abs().max() < 1e-6
    tab = rnd(7, 512).requires_grad_(True)
    idx = torch.randint(0, 7, (3, 11), generator=g)
    (torch.nn.functional.embedding(idx, tab) * 22.627416997969522).backward(ge := rnd(3, 11, 512))
    td = tab.detach().float().cuda().requires_grad_(True)
    ed = ops.embedding(idx.cuda(), td, 22.627416997969522)
    ed.backward(ge.float().cuda())
    assert (td.grad.cpu().double() - tab.grad).abs().max() < 1e-4
    lg = rnd(20, 7).requires_grad_(True)
    rows = torch.tensor([0, 1, 2, 7, 8, 14, 19])
    tg = torch.randint(0, 7, (7,), generator=g)
    loss = torch.nn.functional.cross_entropy(lg[rows], tg)
    loss.backward()
    ld = lg.detach().float().cuda().requires_grad_(True)
    lossd = ops.cross_entropy(ops.gather_rows(ld, rows.cuda()), tg.cuda())
    lossd.backward()
    assert abs(lossd.item() - loss.item()) < 1e-6 and (ld.grad.cpu().double() - lg.grad).abs().max() < 1e-6
    # Adadelta, two steps
    p0, g0 = rnd(1000).float(), rnd(1000).float()
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adadelta([pt], lr=1.0, rho=0.9)
    pdv, sq, acc = p0.clone().cuda(), torch.zeros(1000, device="cuda"), torch.zeros(1000, device="cuda")
    for _ in range(2):
        pt.grad = g0.clone()
        opt.step()
        ops.adadelta(pdv, g0.cuda(), sq, acc, 1.0, 0.9, 1e-6)
    assert (pdv.cpu() - pt.detach()).abs().max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 3], ids=["bf16x3", "dgrad16"])
def test_sld_step_golden(golden_dir, mode):
    """the HIP product model under train.py's step against fixture F7 (reference-generated): schema, ragged
    predictions and cross-entropy within 1e-3, attention map, encoder features, gradients, BatchNorm running stats,
    and the two-step Adadelta trajectory through the engine"""
    from fudanocr_amd import _lib
    from fudanocr_amd.sld import util
    from fudanocr_amd.sld.engine import SLDTrainStep
    g, sc, traj, gn = _fx(golden_dir)
    old = _lib.get_precision()
    _lib.set_precision(mode)
    try:
        m = _build_gpu()
        assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()] == sc["schema"]
        image, _ = make_sld_batch(4, 1234)
        length, text_input, text_gt, _ = util.converter("stroke", sc["labels"], sc["table"])
        assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
        step = SLDTrainStep(m, dropout=False)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.eval()
        res = m(image.cuda(), length, text_input)
        assert _rel(res["pred"].detach().cpu(), g["pred"]) < 1e-3
        assert _rel(res["conv"].detach().cpu()[:, ::64, ::4, ::4], g["conv_sub"]) < 1e-3
        assert _rel(res["map"].detach().cpu()[:, :, :, ::16], g["map_sub"]) < 1e-3
        m.load_state_dict({k: v for k, v in _build_gpu().state_dict().items()})     # undo the BN running-stat update
        out = step(image.cuda(), length, text_input, text_gt)
        assert abs(out["loss"].item() - float(g["loss"])) < 1e-3 * float(g["loss"])
        assert _rel(m.encoder.bn1.running_mean.cpu(), g["bn1_rm"]) < 1e-3
        P = dict(m.named_parameters())
        top = max(v for v in gn.values() if v is not None)
        bad = []
        for k, v in gn.items():
            got = P[k].grad
            if v is None:
                if got is not None and float(got.abs().max()) != 0.0:
                    bad.append((k, "dead parameter has a gradient"))
            else:
                gv = float(got.norm())
                if not abs(gv - v) <= 2e-2 * v + 1e-4 * top:
                    bad.append((k, gv, v))
        assert not bad, bad[:10]
        assert _rel(P["generator_word.proj.weight"].grad.cpu(), g["g_gen_w"]) < 2e-2
        assert _rel(P["embedding_word.lut.weight"].grad.cpu(), g["g_emb"]) < 2e-2
        out2 = step(image.cuda(), length, text_input, text_gt)
        assert abs(out2["loss"].item() - traj["loss"][1]) < 5e-2 * traj["loss"][1]
    finally:
        _lib.set_precision(old)


@pytest.mark.gpu
def test_sld_full_size_properties():
    """BASELINE configs[4] at its per-GPU size (batch 32): size-independent checks of the SLD train step -- finite,
    bounded cross-entropy over a few Adadelta steps on a fixed batch (first loss ~ ln 7), and a deterministic forward (two freshly built
    engines agree bit for bit on the first loss with dropout off)."""
    from fudanocr_amd.sld import util
    from fudanocr_amd.sld.engine import SLDTrainStep
    image, labels = make_sld_batch(32, 7)
    length, text_input, text_gt, _ = util.converter("stroke", labels, device="cuda", strokes=True)
    image = image.cuda()
    step = SLDTrainStep(_build_gpu(), dropout=True)
    losses = [step(image, length, text_input, text_gt)["loss"].item() for _ in range(5)]
    # Adadelta at lr 1.0 on name-keyed weights RAISES the loss over the first steps in the reference too (fixture
    # sld_traj2.json: 2.07 -> 4.50), so "decreasing" is not a property of this configuration: finite and bounded is
    assert all(np.isfinite(losses)) and max(losses) < 50 and abs(losses[0] - 2.0) < 0.5, losses
    first = [SLDTrainStep(_build_gpu(), dropout=False)(image, length, text_input, text_gt)["loss"].item() for _ in range(2)]
    assert first[0] == first[1], first


@pytest.mark.gpu
def test_sld_full_size_step_vs_oracle():
    """BASELINE configs[4] at its per-GPU size (batch 32) against the CPU oracle's step (train.py:63-77 restated in
    oracle.sld_oracle), bench arithmetic mode (3), dropout slots in eval: ragged predictions <= 1e-3 of max,
    cross-entropy <= 1e-3, encoder features and attention maps <= 1e-3, gradient norm <= 2e-2."""
    from fudanocr_amd import _lib
    from fudanocr_amd.sld import util
    from fudanocr_amd.sld.engine import SLDTrainStep
    from oracle import sld_oracle as O
    old = _lib.get_precision()
    _lib.set_precision(3)
    nthr = torch.get_num_threads()
    try:
        image, labels = make_sld_batch(32, 2025)
        length, text_input, text_gt, _ = util.converter("stroke", labels, device="cuda", strokes=True)
        m = _build_gpu()
        step = SLDTrainStep(m, dropout=False)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.eval()
        res = m(image.cuda(), length, text_input)
        pred, conv, amap = res["pred"].detach().cpu(), res["conv"].detach().cpu(), res["map"].detach().cpu()
        m.load_state_dict({k: v for k, v in _build_gpu().state_dict().items()})     # undo the BN running-stat update
        out = step(image.cuda(), length, text_input, text_gt)
        gn_hip = float(step.flat.flat_grad.double().norm())      # Adadelta does not touch the gradient buffer
        P = O.make_params()
        fill_dict_({k: v.data for k, v in P.items()})
        opt = O.AdadeltaState([v for k, v in P.items() if v.requires_grad])
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        r = O.train_step(P, opt, image, length.cpu(), text_input.cpu(), text_gt.cpu())
        e = (_rel(pred, r["pred"]), abs(out["loss"].item() - r["loss"]) / r["loss"], _rel(conv, r["conv"]),
             _rel(amap, r["map"]), abs(gn_hip - r["grad_norm"]) / r["grad_norm"])
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(d):
            with open(os.path.join(d, "test_margins.txt"), "a") as f:
                f.write("sld_full_size_step_vs_oracle (B = 32, mode 3): pred %.2e loss %.2e conv %.2e map %.2e grad-norm %.2e\n" % e)
        assert e[0] < 1e-3 and e[1] < 1e-3 and e[2] < 1e-3 and e[3] < 1e-3, e
        assert e[4] < 2e-2, e
    finally:
        torch.set_num_threads(nthr)
        _lib.set_precision(old)

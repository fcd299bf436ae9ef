"""Row N1: the text-focus loss (frozen transformer recognizer + three-term criterion).
CPU: the oracle restatement against the reference-generated fixture (tools/make_golden_tfl.py).
GPU (-m gpu): the HIP product (`fudanocr_amd.loss.text_focus_loss.TextFocusLoss`) against fixture and oracle."""
import json
import os

import numpy as np
import pytest
import torch

from fudanocr_amd.utils.synth import make_batch
from fudanocr_amd.utils.weight_fill import fill_dict_


def _rel(got, ref):
    got, ref = torch.as_tensor(np.asarray(got)).double(), torch.as_tensor(np.asarray(ref)).double()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def make_sr(hr, seed=7):
    g = torch.Generator().manual_seed(seed)
    noise = (torch.randint(0, 1 << 24, hr.shape, generator=g).to(torch.float32) / float(1 << 24) - 0.5) * 0.25
    return (hr + noise).clamp(0, 1)


def _rel_q(got, ref, q=0.98):
    """q-quantile of |got - ref| over max |ref|.  The fixture images have flat (clamped 0 / 1) regions, where the 2x2
    max-pool windows of the recognizer hold mathematically EQUAL values: which element wins -- and so where the gradient
    is routed -- is decided by the last bit of the convolution's summation order.  Such a flip changes the gradient in
    one receptive field (seen: one ~10 x 10 px patch of one sample, identical in fp32-MFMA and split-bf16 modes); the
    quantile bounds everything outside such patches tightly, `_rel` bounds the patches loosely."""
    got, ref = torch.as_tensor(np.asarray(got)).double(), torch.as_tensor(np.asarray(ref)).double()
    return (torch.quantile((got - ref).abs().flatten(), q) / ref.abs().max()).item()


def map_probe(shape, seed=99):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 1 << 16, tuple(shape), generator=g).to(torch.float32) / float(1 << 16) - 0.5


def test_tfl_oracle_matches_reference_fixture(golden_dir):
    from oracle import tfl_oracle as O
    g = np.load(os.path.join(golden_dir, "tfl_step.npz"))
    sc = json.load(open(os.path.join(golden_dir, "tfl_schema.json")))
    P = O.make_params()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in P.items()] == sc["schema"]
    fill_dict_(P)
    _, hr, labels = make_batch(4, 1234)
    assert labels == sc["labels"]
    length, text_input, text_gt = O.label_encoder(sc["encoded"])
    assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
    sr = make_sr(hr).requires_grad_(True)
    table = torch.tensor(g["table"])
    loss, mse, att, rec, pred, amap, conv = O.text_focus_loss(P, sr, hr, sc["encoded"], table)
    for got, want in zip((loss, mse, att, rec), g["losses"]):
        assert abs(got.item() - want) <= 1e-4 * abs(want) + 1e-12, (got.item(), want)
    assert _rel(pred.detach(), g["pred"]) < 1e-4
    assert _rel(amap.detach()[:, ::4, :, ::8], g["map_sub"]) < 1e-4
    assert _rel(conv.detach()[:, ::64, ::2, ::4], g["conv_sub"]) < 1e-4
    d_rec, = torch.autograd.grad(att * 10 + rec * 0.0005, sr, retain_graph=True)
    assert _rel(d_rec[:, :, ::2, ::4], g["dsr_rec_sub"]) < 1e-3
    d_ce, = torch.autograd.grad(rec, sr, retain_graph=True)
    assert _rel(d_ce[:, :, ::2, ::4], g["dsr_ce_sub"]) < 1e-4
    d_map, = torch.autograd.grad((amap * map_probe(amap.shape)).sum(), sr, retain_graph=True)
    assert _rel(d_map[:, :, ::2, ::4], g["dsr_map_sub"]) < 1e-4
    loss.backward()
    assert _rel(sr.grad[:, :, ::2, ::4], g["dsr_sub"]) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 3], ids=["bf16x3", "dgrad16"])
def test_text_focus_loss_golden(golden_dir, mode):
    """HIP TextFocusLoss (frozen recognizer, eval-mode BatchNorm, 16 x 64 attention, L1 on the attention maps, weighted
    cross-entropy) against the reference-generated fixture: the four loss values, predictions, attention map, encoder
    features, and d loss / d SR image (total and recognizer part alone)"""
    import types
    from fudanocr_amd import _lib
    from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
    from fudanocr_amd.loss.transformer import Transformer
    from fudanocr_amd.utils.weight_fill import fill_module_
    g = np.load(os.path.join(golden_dir, "tfl_step.npz"))
    sc = json.load(open(os.path.join(golden_dir, "tfl_schema.json")))
    old = _lib.get_precision()
    _lib.set_precision(mode)
    try:
        tr = fill_module_(Transformer())
        assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in tr.state_dict().items()] == sc["schema"]
        tr = tr.cuda().eval()
        for p in tr.parameters():
            p.requires_grad = False
        crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, weight_table=torch.tensor(g["table"]))
        _, hr, labels = make_batch(4, 1234)
        sr = make_sr(hr).cuda().requires_grad_(True)
        loss, mse, att, rec = crit(sr, hr.cuda(), labels)
        for got, want, tol in zip((loss, mse, att, rec), g["losses"], (1e-3, 1e-3, 2e-2, 1e-3)):
            assert abs(got.item() - want) <= tol * abs(want), (got.item(), want)
        enc = [s.lower() + "-" for s in labels]
        length, text_input, text_gt = crit.label_encoder(enc)
        assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
        from fudanocr_amd.loss.text_focus_loss import to_gray_tensor
        with torch.no_grad():
            gray = to_gray_tensor(sr.detach())
            ref_gray = 0.299 * sr[:, 0:1] + 0.587 * sr[:, 1:2] + 0.114 * sr[:, 2:3]
            assert (gray - ref_gray).abs().max() < 1e-6
            pred, amap, _ = tr(gray, length, text_input)
            conv = tr.encoder(gray)
        assert _rel(pred.cpu(), g["pred"]) < 1e-3
        assert _rel(amap.cpu()[:, ::4, :, ::8], g["map_sub"]) < 1e-3
        assert _rel(conv.permute(0, 3, 1, 2).cpu()[:, ::64, ::2, ::4], g["conv_sub"]) < 1e-3
        # smooth probes pin the recognizer's data gradient: cross-entropy alone, and a linear functional of the map
        d_ce, = torch.autograd.grad(rec, sr, retain_graph=True)
        assert _rel_q(d_ce.cpu()[:, :, ::2, ::4], g["dsr_ce_sub"]) < (2e-2 if mode == 3 else 5e-3)
        assert _rel(d_ce.cpu()[:, :, ::2, ::4], g["dsr_ce_sub"]) < 0.15
        sr2 = sr.detach().clone().requires_grad_(True)
        _, amap2, _ = tr(to_gray_tensor(sr2), length, text_input)
        d_map, = torch.autograd.grad((amap2 * map_probe(amap2.shape).cuda()).sum(), sr2)
        assert _rel_q(d_map.cpu()[:, :, ::2, ::4], g["dsr_map_sub"]) < (2e-2 if mode == 3 else 5e-3)
        assert _rel(d_map.cpu()[:, :, ::2, ::4], g["dsr_map_sub"]) < 0.15
        # the L1 term's gradient is sign(map_sr - map_hr) / N with |map_sr - map_hr| ~ 1e-8 under these weights: signs of
        # near-ties flip with the rounding of the arithmetic mode, so this one is a loose sanity bound only
        d_rec, = torch.autograd.grad(att * 10 + rec * 0.0005, sr, retain_graph=True)
        assert _rel(d_rec.cpu()[:, :, ::2, ::4], g["dsr_rec_sub"]) < 0.25
        loss.backward()
        assert _rel(sr.grad.cpu()[:, :, ::2, ::4], g["dsr_sub"]) < 1e-3
        # the sentinel branch (text_focus off) keeps the reference's 4-tuple shape
        off = TextFocusLoss(types.SimpleNamespace(text_focus=False))(sr.detach(), hr.cuda(), labels)
        assert off[2] == -1 and off[3] == -1 and abs(off[0].item() - g["losses"][1]) < 1e-3 * g["losses"][1]
    finally:
        _lib.set_precision(old)


@pytest.mark.gpu
def test_l1_and_weight_cross_entropy_kernels():
    """focr_l1_fwd/bwd and focr_weight_cross_entropy_fwd against the torch formulas (L1Loss; loss/weight_ce_loss.py:
    38-45: -log(sum_c softmax_c w[t][c] ... ) restated in tfl_oracle.weight_cross_entropy)"""
    from fudanocr_amd.sld import ops
    from oracle import tfl_oracle as O
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(3, 16, 7, 256, generator=g), torch.rand(3, 16, 7, 256, generator=g)
    bc = b.cuda().requires_grad_(True)
    l = ops.l1_loss(a.cuda(), bc)
    (l * 3).backward()
    br = b.clone().requires_grad_(True)
    lr = torch.nn.functional.l1_loss(a, br)
    (lr * 3).backward()
    assert abs(l.item() - lr.item()) < 1e-6 and torch.equal(bc.grad.cpu(), br.grad)
    logits = torch.randn(23, 37, generator=g) * 3
    tgt = torch.randint(0, 37, (23,), generator=g)
    table = torch.rand(37, 37, generator=g) + 0.5
    lc = logits.cuda().requires_grad_(True)
    w = ops.weight_cross_entropy(lc, tgt.cuda(), table.cuda())
    (w * 2).backward()
    lref = logits.clone().requires_grad_(True)
    wr = O.weight_cross_entropy(lref, tgt, table)
    (wr * 2).backward()
    assert abs(w.item() - wr.item()) < 1e-5 * abs(wr.item()) and _rel(lc.grad.cpu(), lref.grad) < 1e-5


@pytest.mark.gpu
def test_harness_trains_with_text_focus(tmp_path, monkeypatch):
    """main.py --text_focus: the harness builds TextFocusLoss as the criterion (reference interfaces/base.py:143-150)
    and the engine steps through it (gradient reaches the SR network through the frozen recognizer)"""
    import yaml
    from fudanocr_amd import main as M
    from fudanocr_amd.utils.util import AttrDict
    monkeypatch.chdir(tmp_path)
    cfg = AttrDict(yaml.load(open(os.path.join(os.path.dirname(M.__file__), "config", "super_resolution.yaml")),
                             Loader=yaml.Loader))
    cfg.TRAIN.iters_per_epoch, cfg.TRAIN.displayInterval, cfg.TRAIN.saveInterval = 2, 1, 100
    res = M.main(cfg, M.parse(["--arch", "tbsrn", "--STN", "--exp_name", "tf", "--batch_size", "4", "--text_focus",
                              "--standin_assets"]))
    assert res["images_per_sec"] > 0
    assert set(cfg["standin_assets"]) == {"pretrain_transformer.pth", "confuse.pkl"}
    # without the explicit opt-in the missing assets are an error, as in the reference (text_focus_loss.py:56-58)
    with pytest.raises(FileNotFoundError):
        M.main(cfg, M.parse(["--arch", "tbsrn", "--STN", "--exp_name", "tf2", "--batch_size", "4", "--text_focus"]))


@pytest.mark.gpu
def test_trainstep_text_focus_gradient_is_loss_times_100():
    """engine.TrainStep with TextFocusLoss (loss = mse + 10 * attention + 0.0005 * recognition, reference
    loss/text_focus_loss.py:84-99; step tail interfaces/super_resolution.py:79-82): the flat gradient the engine
    produces equals plain autograd's (loss * 100).backward() on a twin network.  The engine's 'feed 100 into mse and the
    recognition term' shortcut is only valid for CTCFocusLoss (loss = mse + ctc) and must not be taken here -- forcing it
    gives a different gradient, which this test also demonstrates."""
    import types
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
    from fudanocr_amd.loss.transformer import Transformer
    from fudanocr_amd.model import tbsrn
    from fudanocr_amd.utils.weight_fill import fill_module_
    _lib.load()
    dev = torch.device("cuda", 0)
    tr = fill_module_(Transformer()).to(dev).eval()
    for p in tr.parameters():
        p.requires_grad = False
    lr_img, hr, labels = make_batch(4, 1234)
    lr_img, hr = lr_img.to(dev), hr.to(dev)

    table = torch.rand(37, 37, generator=torch.Generator().manual_seed(3)) + 0.5      # stands in for confuse.pkl

    def crit_():
        return TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, weight_table=table)

    def flat_grad(force_shortcut):
        net = fill_module_(tbsrn.TBSRN(STN=True)).to(dev)
        crit = crit_()
        assert not getattr(crit, "LOSS_IS_MSE_PLUS_REC", False)
        if force_shortcut:
            crit.LOSS_IS_MSE_PLUS_REC = True
        step = TrainStep(net, crit, lr=0.0, dropout=False)
        out = step(lr_img, hr, labels)
        torch.cuda.synchronize()
        names = [n for n, p in net.named_parameters() if p.requires_grad]
        return {n: dict(net.named_parameters())[n].grad.detach().clone() for n in names}, out

    g_engine, out = flat_grad(False)
    # twin network, no engine: torch autograd end to end
    net = fill_module_(tbsrn.TBSRN(STN=True)).to(dev).train()
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    crit = crit_()
    loss, mse, att, rec = crit(net(lr_img), hr, labels)
    assert abs(loss.item() - out["loss"].item()) <= 1e-6 * abs(loss.item())
    assert abs(loss.item() - (mse + 10 * att + 0.0005 * rec).item()) <= 1e-5 * abs(loss.item())
    (loss * 100).backward()
    torch.cuda.synchronize()
    num, den, per = 0.0, 0.0, []
    for n, p in net.named_parameters():
        if p.grad is None:
            assert float(g_engine[n].abs().max()) == 0.0, n
            continue
        d = (p.grad - g_engine[n]).double()
        num += float((d * d).sum())
        den += float((p.grad.double() ** 2).sum())
        per.append((n, float(d.norm()), float(p.grad.double().norm())))
    assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5
    # per parameter too (biases in front of a train-mode BatchNorm have a mathematically zero gradient: rounding noise of
    # either side, bounded against the whole gradient's norm instead of their own)
    bad = [(n, dn, gn) for n, dn, gn in per if dn > 1e-4 * gn + 1e-7 * den ** 0.5]
    assert not bad, bad[:5]
    g_short, _ = flat_grad(True)
    n0 = "block1.0.weight"
    diff = float((g_short[n0] - g_engine[n0]).norm()) / float(g_engine[n0].norm())
    # (with name-keyed recognizer weights the recognizer-path gradient is small beside the MSE's: the wrong weighting is
    # visible, not dominant)
    assert diff > 1e-6, "the shortcut would have been indistinguishable (%g): test is vacuous" % diff


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["tfl", "sfl"])
def test_text_focus_full_size_step_vs_oracle(kind):
    """One whole optimisation step of TBSRN under the reference's REAL criteria at B = 32 -- TextFocusLoss
    (loss/text_focus_loss.py:84-99, frozen recognizer loss/transformer.py:82-389 on HR and SR) and text-gestalt's
    StrokeFocusLoss (loss/stroke_focus_loss.py:83-118) -- HIP engine against the CPU oracle (oracle/tfl_oracle.py
    train_step_focus) on the same seeded batch and name-keyed weights: SR pixels, every loss term, the pre-clip gradient
    norm, and the updated parameters."""
    import types
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.loss.stroke_focus_loss import StrokeFocusLoss, standin_decomposition
    from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
    from fudanocr_amd.model import tbsrn
    from fudanocr_amd.utils.weight_fill import fill_module_
    from oracle import sr_oracle as SO
    from oracle import tfl_oracle as O
    _lib.load()
    dev = torch.device("cuda", 0)
    B = 32
    lr_img, hr, labels = make_batch(B, 4321)
    table = torch.rand(37, 37, generator=torch.Generator().manual_seed(3)) + 0.5
    dic = standin_decomposition()
    if kind == "tfl":
        from fudanocr_amd.loss.transformer import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr, weight_table=table)
        R = O.make_params()
    else:
        from fudanocr_amd.loss.transformer_english_decomposition import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = StrokeFocusLoss(types.SimpleNamespace(text_focus=True, stroke_lambda=50), transformer=tr, decomposition=dic)
        R = O.make_stroke_params()
    for p in tr.parameters():
        p.requires_grad = False
    net = fill_module_(tbsrn.TBSRN(STN=True)).to(dev)
    step = TrainStep(net, crit, dropout=False)
    out = step(lr_img.to(dev), hr.to(dev), labels)
    torch.cuda.synchronize()
    # ---- oracle
    fill_dict_(R)
    P = SO.make_params(SO.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    opt = SO.AdamState([v for v in P.values() if v.requires_grad])
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        ref = O.train_step_focus(P, opt, R, lr_img, hr, labels, kind, table, dic, 50.0)
    finally:
        torch.set_num_threads(old)
    sr_err = float((out["sr"].cpu() - ref["sr"]).abs().max() / ref["sr"].abs().max())
    assert sr_err < 1e-3, sr_err
    assert abs(out["loss"].item() - ref["loss"]) <= 1e-3 * abs(ref["loss"]), (out["loss"].item(), ref["loss"])
    assert abs(out["mse"].item() - ref["mse"]) <= 1e-3 * abs(ref["mse"])
    assert abs(step.opt.grad_norm().item() - ref["grad_norm"]) <= 2e-2 * ref["grad_norm"], (step.opt.grad_norm().item(), ref["grad_norm"])
    # the updated parameters (Adam's normalised step: +-lr per element wherever the gradient's sign is above the noise)
    sd = net.state_dict()
    worst = 0.0
    for k in ("block1.0.weight", "block2.conv1.weight", "block4.feature_enhancer.linear.weight", "block7.0.weight"):
        d = float((sd[k].detach().cpu() - P[k].detach()).abs().max())
        worst = max(worst, d)
    assert worst <= 2.5e-4, worst


# ---------------------------------------------------------------------------------------------------------------------
# text-gestalt half of row N1: StrokeFocusLoss + the stroke-level recognizer (fixture tools/make_golden_sfl.py)
# ---------------------------------------------------------------------------------------------------------------------
def test_stroke_oracle_matches_reference_fixture(golden_dir):
    from oracle import tfl_oracle as O
    g = np.load(os.path.join(golden_dir, "sfl_step.npz"))
    sc = json.load(open(os.path.join(golden_dir, "sfl_schema.json")))
    P = O.make_stroke_params()
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in P.items()] == sc["schema"]
    assert "embedding_word_with_upperword.lut.weight" in P and P["generator_word_with_upperword.proj.weight"].shape[0] == 10
    fill_dict_(P)
    _, hr, _ = make_batch(4, 1234)
    labels, dic = sc["labels"], sc["decomposition"]
    assert "#" in labels[1] and "#" not in dic            # a character without a decomposition is skipped
    length, text_input, text_gt = O.label_stroke_encoder(labels, dic)
    assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
    sr = make_sr(hr).requires_grad_(True)
    loss, mse, att, rec, pred, amap, c_hr, c_sr = O.stroke_focus_loss(P, sr, hr, labels, dic, sc["stroke_lambda"])
    assert rec == -1 and g["losses"][3] == -1
    for got, want in zip((loss, mse, att), g["losses"]):
        assert abs(got.item() - want) <= 1e-4 * abs(want) + 1e-12, (got.item(), want)
    assert c_hr == sc["correct_hr"] and c_sr == sc["correct_sr"]
    assert _rel(pred.detach(), g["pred"]) < 1e-4
    assert _rel(amap.detach()[:, ::4, :, ::8], g["map_sub"]) < 1e-4
    d_map, = torch.autograd.grad((amap * map_probe(amap.shape)).sum(), sr, retain_graph=True)
    assert _rel(d_map[:, :, ::2, ::4], g["dsr_map_sub"]) < 1e-4
    loss.backward()
    assert _rel(sr.grad[:, :, ::2, ::4], g["dsr_sub"]) < 1e-4
    with torch.no_grad():
        rgbm = torch.cat([sr.detach(), torch.ones_like(sr[:, :1])], 1)
        pred4, _, _ = O.stroke_recognizer(P, rgbm, length, text_input)
        assert _rel(pred4, g["pred4"]) < 1e-4
        _, _, c_mixed = O.stroke_recognizer(P, O.to_gray(sr.detach()), length, torch.tensor(sc["text_input_mixed"]))
    assert c_mixed == sc["correct_mixed"] and any(c_mixed) and not all(c_mixed)


def test_missing_focus_assets_raise_unless_opted_in(tmp_path, monkeypatch):
    """reference stroke_focus_loss.py:31,45 / text_focus_loss.py:56-58 / weight_ce_loss.py:35: a missing decomposition
    table, recognizer checkpoint or confusion matrix is an error; the stand-ins need an explicit opt-in and are recorded"""
    import types
    from fudanocr_amd.loss import stroke_focus_loss as S
    from fudanocr_amd.loss import text_focus_loss as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("FOCR_ALLOW_STANDIN_ASSETS", raising=False)
    with pytest.raises(FileNotFoundError, match="english_decomposition.txt"):
        S.load_decomposition()
    with pytest.raises(FileNotFoundError, match="confuse.pkl"):
        T.load_confuse_matrix()
    with pytest.raises(FileNotFoundError):
        S.StrokeFocusLoss(types.SimpleNamespace(text_focus=False), device="cpu")
    used = []
    assert len(S.load_decomposition(allow_standin=True, used=used)) == 62 and used == ["english_decomposition.txt"]
    assert T.load_confuse_matrix(allow_standin=True, used=used).shape == (37, 37) and used[-1] == "confuse.pkl"
    monkeypatch.setenv("FOCR_ALLOW_STANDIN_ASSETS", "1")
    crit = S.StrokeFocusLoss(types.SimpleNamespace(text_focus=False), device="cpu")
    assert crit.standin_assets == ["english_decomposition.txt"]
    # a real table on disk is used as it is and nothing is recorded
    os.makedirs(tmp_path / "dataset" / "mydata")
    (tmp_path / "dataset" / "mydata" / "english_decomposition.txt").write_text("a 12\nb 3\n")
    monkeypatch.delenv("FOCR_ALLOW_STANDIN_ASSETS")
    crit = S.StrokeFocusLoss(types.SimpleNamespace(text_focus=False), device="cpu")
    assert crit.dic == {"a": "12", "b": "3"} and crit.standin_assets == []


def test_stroke_standin_table_and_encoder_host_logic():
    """the stand-in decomposition has the file's format (digits 1-9 per character) and is the table the fixture used"""
    from fudanocr_amd.loss.stroke_focus_loss import standin_decomposition
    sc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sfl_schema.json")))
    dic = standin_decomposition()
    assert dic == sc["decomposition"]
    assert all(1 <= len(v) <= 4 and set(v) <= set("123456789") for v in dic.values()) and len(dic) == 62


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 3], ids=["bf16x3", "dgrad16"])
def test_stroke_focus_loss_golden(golden_dir, mode):
    """HIP StrokeFocusLoss (text-gestalt): stroke encoding, 10-class recognizer with the reference's key names, correct_list,
    loss = mse + stroke_lambda * L1(attention maps), d loss / d SR -- against the reference-generated fixture"""
    import types
    from fudanocr_amd import _lib
    from fudanocr_amd.loss.stroke_focus_loss import StrokeFocusLoss
    from fudanocr_amd.loss.text_focus_loss import to_gray_tensor
    from fudanocr_amd.loss.transformer_english_decomposition import Transformer
    from fudanocr_amd.utils.weight_fill import fill_module_
    g = np.load(os.path.join(golden_dir, "sfl_step.npz"))
    sc = json.load(open(os.path.join(golden_dir, "sfl_schema.json")))
    old = _lib.get_precision()
    _lib.set_precision(mode)
    try:
        tr = fill_module_(Transformer())
        assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in tr.state_dict().items()] == sc["schema"]
        tr = tr.cuda().eval()
        for p in tr.parameters():
            p.requires_grad = False
        args = types.SimpleNamespace(text_focus=True, stroke_lambda=sc["stroke_lambda"])
        crit = StrokeFocusLoss(args, transformer=tr, decomposition=sc["decomposition"])
        _, hr, _ = make_batch(4, 1234)
        labels = sc["labels"]
        length, text_input, text_gt = crit.label_stroke_encoder(labels)
        assert length.tolist() == sc["length"] and text_input.tolist() == sc["text_input"] and text_gt.tolist() == sc["text_gt"]
        sr = make_sr(hr).cuda().requires_grad_(True)
        loss, mse, att, rec = crit(sr, hr.cuda(), labels)
        assert rec == -1
        for got, want, tol in zip((loss, mse, att), g["losses"], (1e-3, 1e-3, 2e-2)):
            assert abs(got.item() - want) <= tol * abs(want), (got.item(), want)
        with torch.no_grad():
            gray = to_gray_tensor(sr.detach())
            pred, amap, c_sr = tr(gray, length, text_input)
            _, _, c_hr = tr(to_gray_tensor(hr.cuda()), length, text_input)
            pred4, _, _ = tr(torch.cat([sr.detach(), torch.ones_like(sr[:, :1])], 1), length, text_input)
            _, _, c_mixed = tr(gray, length, torch.tensor(sc["text_input_mixed"]).cuda())
            assert tr(gray, length, text_input, want_correct=False)[2] is None
        assert c_sr == sc["correct_sr"] and c_hr == sc["correct_hr"] and c_mixed == sc["correct_mixed"]
        assert _rel(pred.cpu(), g["pred"]) < 1e-3 and _rel(pred4.cpu(), g["pred4"]) < 1e-3
        assert _rel(amap.cpu()[:, ::4, :, ::8], g["map_sub"]) < 1e-3
        sr2 = sr.detach().clone().requires_grad_(True)
        _, amap2, _ = tr(to_gray_tensor(sr2), length, text_input, want_correct=False)
        d_map, = torch.autograd.grad((amap2 * map_probe(amap2.shape).cuda()).sum(), sr2)
        assert _rel_q(d_map.cpu()[:, :, ::2, ::4], g["dsr_map_sub"]) < (2e-2 if mode == 3 else 5e-3)
        assert _rel(d_map.cpu()[:, :, ::2, ::4], g["dsr_map_sub"]) < 0.15
        loss.backward()
        # total gradient = MSE part (exact) + 50 * L1 part (sign of ~1e-8 map differences: loose, see the text-focus test)
        assert _rel(sr.grad.cpu()[:, :, ::2, ::4], g["dsr_sub"]) < 2e-2
        off = StrokeFocusLoss(types.SimpleNamespace(text_focus=False), decomposition=sc["decomposition"])(
            sr.detach(), hr.cuda(), labels)
        assert off[2] == -1 and off[3] == -1 and abs(off[0].item() - g["losses"][1]) < 1e-3 * g["losses"][1]
    finally:
        _lib.set_precision(old)


@pytest.mark.gpu
def test_masked_focus_losses_match_the_packed_forms():
    """focr_l1_masked_* / focr_weight_cross_entropy_masked_fwd (padded label layout, the real extents in a device plan) against
    torch on the packed layout the reference uses (text_focus_loss.py:92-93: L1 over [B,16,max(len),256], weighted CE over
    the sum(len) real rows): values and gradients; the padded part gets exactly zero gradient."""
    from fudanocr_amd import _lib
    from fudanocr_amd.loss.padded_labels import PaddedLabels
    from fudanocr_amd.sld import ops
    _lib.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(5)
    ids = [[3, 7, 1, 0], [5, 0], [9, 9, 2, 4, 6, 1, 0], [8, 0], [1, 2, 3, 0]]
    B, C = len(ids), 37
    for bucket in (0, 8, 16):
        enc = PaddedLabels.build(ids, dev, bucket)
        L, lmax, total = enc.cap, 7, sum(len(s) for s in ids)
        assert (L == 7 if bucket == 0 else L == bucket) and enc.plan[:2].tolist() == [lmax, total]
        assert enc.text_input[2].tolist()[:7] == [0, 9, 9, 2, 4, 6, 1] and enc.text_input[1].tolist()[:3] == [0, 5, 0]
        a = torch.rand(B, 16, L, 256, generator=g).to(dev)
        b = torch.rand(B, 16, L, 256, generator=g).to(dev).requires_grad_(True)
        loss = ops.l1_loss_masked(a, b, enc.plan)
        (loss * 3.0).backward()
        bd = b.detach().double().cpu().requires_grad_(True)
        ref = (a.double().cpu()[:, :, :lmax] - bd[:, :, :lmax]).abs().mean()
        (ref * 3.0).backward()
        assert abs(loss.item() - ref.item()) <= 1e-6 * ref.item()
        assert float((b.grad.double().cpu() - bd.grad).abs().max()) <= 1e-6 * float(bd.grad.abs().max())
        assert float(b.grad[:, :, lmax:].abs().max()) == 0.0 if L > lmax else True
        # weighted cross entropy: padded rows against the packed op on the gathered rows
        table = (torch.rand(C, C, generator=g) + 0.5).to(dev)
        x = torch.randn(B * L, C, generator=g).to(dev).requires_grad_(True)
        lw = ops.weight_cross_entropy_masked(x, enc.text_gt, table, enc.plan)
        (lw * 2.0).backward()
        rows = [i * L + j for i, s in enumerate(ids) for j in range(len(s))]
        xp = x.detach()[rows].clone().requires_grad_(True)
        lp = ops.weight_cross_entropy(xp, torch.tensor([c for s in ids for c in s], device=dev), table)
        (lp * 2.0).backward()
        assert abs(lw.item() - lp.item()) <= 1e-6 * abs(lp.item())
        assert float((x.grad[rows] - xp.grad).abs().max()) <= 1e-6 * float(xp.grad.abs().max())
        pad = sorted(set(range(B * L)) - set(rows))
        assert float(x.grad[pad].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["tfl", "sfl"])
def test_focus_loss_padding_is_neutral(kind):
    """the criterion on labels padded to a capacity bucket (what a recorded step uses) gives the loss terms and the SR
    gradient of the reference-shaped batch (capacity = longest label): padded decoder positions sit behind the causal mask,
    and the masked losses leave them out"""
    import types
    from fudanocr_amd import _lib
    from fudanocr_amd.loss.stroke_focus_loss import StrokeFocusLoss, standin_decomposition
    from fudanocr_amd.loss.text_focus_loss import TextFocusLoss
    from fudanocr_amd.utils.weight_fill import fill_module_
    _lib.load()
    dev = torch.device("cuda", 0)
    if kind == "tfl":
        from fudanocr_amd.loss.transformer import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = TextFocusLoss(types.SimpleNamespace(text_focus=True), transformer=tr,
                             weight_table=torch.rand(37, 37, generator=torch.Generator().manual_seed(3)) + 0.5)
    else:
        from fudanocr_amd.loss.transformer_english_decomposition import Transformer
        tr = fill_module_(Transformer()).to(dev).eval()
        crit = StrokeFocusLoss(types.SimpleNamespace(text_focus=True, stroke_lambda=50), transformer=tr,
                               decomposition=standin_decomposition())
    for p in tr.parameters():
        p.requires_grad = False
    _, hr, labels = make_batch(6, 99)
    hr = hr.to(dev)
    sr0 = (hr + 0.05 * torch.randn(hr.shape, generator=torch.Generator().manual_seed(1)).to(dev)).clamp(0, 1)
    res = []
    lmax = crit.encode(labels, dev, 0).cap
    for bucket in (0, lmax + 3, 2 * lmax + 5):            # capacity = the longest label, then two larger ones
        enc = crit.encode(labels, dev, bucket)
        sr = sr0.clone().requires_grad_(True)
        loss, mse, att, rec = crit(sr, hr, None, enc)
        loss.backward()
        res.append((enc.cap, loss.item(), att.item(), rec.item() if torch.is_tensor(rec) else 0.0, sr.grad.clone()))
    assert res[0][0] < res[1][0] < res[2][0]
    for cap, l, a, r, gr in res[1:]:
        assert abs(l - res[0][1]) <= 1e-6 * abs(res[0][1]) and abs(a - res[0][2]) <= 2e-6 * abs(res[0][2])
        assert abs(r - res[0][3]) <= 2e-6 * abs(res[0][3]) + 1e-12
        assert float((gr - res[0][4]).abs().max()) <= 1e-5 * float(res[0][4].abs().max())

"""GPU parity tests, model level: the HIP product modules (fudanocr_amd.model.*) against
 (a) the golden vectors generated from the imported reference (tests/golden), and
 (b) the CPU oracle on fresh seeded inputs.
Gate (north_star): SR pixels and losses within 1e-3 relative (pixels: relative to the output's
max magnitude -- near-zero tanh outputs need an absolute floor)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fudanocr_amd.utils.synth import make_batch          # noqa: E402
from fudanocr_amd.utils.weight_fill import fill_module_  # noqa: E402


def rel_to_max(got, ref):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref)).double()
    return ((got - ref).abs().max() / ref.abs().max()).item()


def rel_q(got, ref, q=0.98):
    """q-quantile of |got - ref| over max |ref| (for gradients that pass through max-pool / bilinear-cell decisions on
    inputs with exact ties, e.g. the binary mask channel: a tie decided by the last bit moves the gradient of one
    receptive field; the quantile bounds everything else tightly, rel_to_max bounds the patches loosely)"""
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref)).double()
    return (torch.quantile((got - ref).abs().flatten(), q) / ref.abs().max()).item()


def _note(msg):
    """measured margins of the tolerance tests, kept with the run's other outputs when that directory exists"""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "test_margins.txt"), "a") as f:
            f.write(msg + "\n")


def build(arch="tbsrn"):
    from fudanocr_amd.smoke import build_models
    return build_models(torch.device("cuda:0"), arch)


@pytest.fixture(params=[2, 1, 3], ids=["fastgrad", "bf16x3", "dgrad16"])
def prec_mode(request):
    """model-level parity under the bf16x3 modes: 2 = the default (single-bf16 gradient accumulations in the
    attention backward), 1 = split products everywhere, 3 = 2 + single-bf16 data-gradient convolutions on the halo
    kernel.  Forward results are identical in the three modes."""
    from fudanocr_amd import _lib
    old = _lib.get_precision()
    _lib.set_precision(request.param)
    yield request.param
    _lib.set_precision(old)


def skip_dup(arch, prec_mode):
    if arch == "tsrn" and prec_mode == 1:
        pytest.skip("TSRN has no attention: mode 1 == mode 2")


def eval_dropout(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()


def test_state_dict_schema(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "schema.json")))
    net, rec, _ = build()
    tnet, _, _ = build("tsrn")
    for name, m in (("tbsrn", net), ("crnn", rec), ("tsrn", tnet)):
        mine = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
        assert mine == ref[name]


@pytest.mark.parametrize("arch", ["tbsrn", "tsrn"])
def test_eval_golden(arch, golden_dir):
    g = np.load(os.path.join(golden_dir, "%s_eval.npz" % arch))
    net, _, _ = build(arch)
    net.eval()
    lr, _, _ = make_batch(4, 1234)
    with torch.no_grad():
        sr = net(lr.cuda())
    assert rel_to_max(sr, g["sr"]) < 1e-3


@pytest.mark.parametrize("arch", ["tbsrn", "tsrn"])
def test_train_mse_golden(arch, golden_dir, prec_mode):
    skip_dup(arch, prec_mode)
    g = np.load(os.path.join(golden_dir, "%s_train_mse.npz" % arch))
    gn = json.load(open(os.path.join(golden_dir, "%s_train_mse_gradnorms.json" % arch)))
    net, _, _ = build(arch)
    net.train()
    eval_dropout(net)
    lr, hr, _ = make_batch(4, 1234)
    from fudanocr_amd import kernels as K
    x = lr.cuda().requires_grad_(True)
    sr = net(x)
    mse = K.mse_loss(sr, hr.cuda())
    (mse * 100).backward()
    assert rel_to_max(sr, g["sr"]) < 1e-3
    assert abs(mse.item() - float(g["mse"])) < 1e-3 * float(g["mse"])
    # d loss / d LR image (through the TPS sampling AND the STN head): fixture F5's `dlr`
    assert rel_to_max(x.grad, g["dlr"]) < 2e-2
    P = dict(net.named_parameters())
    assert rel_to_max(P["block1.0.weight"].grad, g["g_block1_w"]) < 2e-2
    assert rel_to_max(P["block8.1.bias"].grad, g["g_block8_b"]) < 2e-2
    # fc2.weight.grad = sum_b dctrl[b]^T feat[b] cancels heavily across samples: one bilinear-cell
    # flip (fp32 rounding of a sampling coordinate) moves it by ~10 % of its max while d ctrl itself
    # agrees to <1 % (tools/diag_stn_grad.py); gate it loosely here, tightly through the norms below.
    # (round 6: measured max 1.2e-2 .. 1.8e-2, 98 % quantile 2.6e-3 .. 5.4e-3, 90 % quantile 5e-4 .. 2.8e-3 over both
    # architectures and modes 1 / 2 / 3 -- profiles/r06_test_margins.txt; the gates leave 3 x for a cell flip)
    assert rel_to_max(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"]) < 6e-2
    assert rel_q(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"], 0.98) < 1.5e-2
    assert rel_q(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"], 0.90) < 8e-3
    _note("stn_fc2.weight grad vs fixture (%s, mode %d): max %.3e, q98 %.3e, q90 %.3e, median %.3e" % (
        arch, prec_mode, rel_to_max(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"]),
        rel_q(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"], 0.98), rel_q(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"], 0.9),
        rel_q(P["stn_head.stn_fc2.weight"].grad, g["g_fc2_w"], 0.5)))
    assert rel_to_max(P["block2.conv1.weight"].grad, g["g_b2c1_w"]) < 2e-2
    sd = net.state_dict()
    assert rel_to_max(sd["block2.bn1.running_mean"], g["bn_rm"]) < 1e-3
    assert rel_to_max(sd["block2.bn1.running_var"], g["bn_rv"]) < 1e-3
    top = max(v for v in gn.values() if v is not None)
    bad = []
    for k, v in gn.items():
        got = P[k].grad
        if v is None:
            if got is not None and float(got.abs().max()) != 0.0:
                bad.append((k, "dead parameter has a gradient"))
        else:
            gv = float(got.norm()) if got is not None else float("nan")
            if not abs(gv - v) <= 2e-2 * v + 1e-4 * top:
                bad.append((k, gv, v))
    assert not bad, bad[:10]


def test_crnn_decoded_strings_golden(golden_dir):
    """rows a21 / N4: CRNN logits computed on the GPU, argmax on the device, greedy decode -> the strings the
    REFERENCE's decoders produced from the reference's logits (fixture decode.json, case crnn_leg)"""
    fx = json.load(open(os.path.join(golden_dir, "decode.json")))
    case = next(c for c in fx["cases"] if c["name"] == "crnn_leg")
    _, rec, _ = build()
    from fudanocr_amd import kernels as K
    from fudanocr_amd.utils.utils_crnn import get_crnn_pred
    img = torch.rand(4, 3, 32, 128, generator=torch.Generator().manual_seed(77))
    with torch.no_grad():
        out = rec(K.bicubic_gray(img.cuda(), 100)).permute(1, 0, 2).contiguous()       # [B, T, 37]
    # name-keyed random weights give flat logits (top-2 margins down to 3e-4): frames whose golden margin is inside the
    # 1e-3 logit tolerance may legitimately flip, every other frame's argmax must match the reference's
    gl = torch.tensor(np.load(os.path.join(golden_dir, "crnn_leg.npz"))["logits"]).permute(1, 0, 2)
    top2 = gl.topk(2, dim=2).values
    decided = (top2[..., 0] - top2[..., 1]) > 2e-3 * gl.abs().max()
    am = out.argmax(2).cpu()
    ref_am = torch.tensor(case["argmax"])
    assert decided.float().mean() > 0.5
    assert torch.equal(am[decided], ref_am[decided])
    patched = torch.nn.functional.one_hot(torch.where(decided, am, ref_am), 37).float().cuda()
    assert get_crnn_pred(patched) == case["get_crnn_pred"]


def test_crnn_leg_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "crnn_leg.npz"))
    _, rec, _ = build()
    from fudanocr_amd import kernels as K
    img = torch.rand(4, 3, 32, 128, generator=torch.Generator().manual_seed(77))
    gray = K.bicubic_gray(img.cuda(), 100)
    assert rel_to_max(gray, g["gray"]) < 1e-5
    with torch.no_grad():
        logits = rec(gray)
    assert rel_to_max(logits, g["logits"]) < 1e-3


@pytest.mark.parametrize("arch", ["tbsrn", "tsrn"])
def test_mask_variant_golden(arch, golden_dir):
    """--mask (in_planes = 4, reference main.py:31; the 9x9 layers with 4 channels run on the generic kernels): schema,
    train-mode forward + MSE backward and eval forward against the reference built with mask=True"""
    from fudanocr_amd import kernels as K
    from fudanocr_amd.model import tbsrn, tsrn
    from fudanocr_amd.utils.synth import with_mask
    from fudanocr_amd.utils.weight_fill import fill_module_
    g = np.load(os.path.join(golden_dir, "%s_mask_train_mse.npz" % arch))
    gn = json.load(open(os.path.join(golden_dir, "%s_mask_gradnorms.json" % arch)))
    schema = json.load(open(os.path.join(golden_dir, "mask_schema.json")))[arch]
    mod = tbsrn.TBSRN if arch == "tbsrn" else tsrn.TSRN
    net = fill_module_(mod(STN=True, mask=True))
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()] == schema
    net = net.cuda().train()
    eval_dropout(net)
    lr, hr, _ = make_batch(4, 1234)
    lr4, hr4 = with_mask(lr), with_mask(hr)
    x = lr4.cuda().requires_grad_(True)
    sr = net(x)
    mse = K.mse_loss(sr, hr4.cuda())
    (mse * 100).backward()
    assert sr.shape == (4, 4, 32, 128) and rel_to_max(sr, g["sr"]) < 1e-3
    assert abs(mse.item() - float(g["mse"])) < 1e-3 * float(g["mse"])
    # d loss / d LR goes through the STN head's max-pools and the TPS bilinear cells; the binary mask channel puts exact
    # ties in front of both
    assert rel_q(x.grad[:, :, ::2, ::4], g["dlr_sub"]) < 2e-2 and rel_to_max(x.grad[:, :, ::2, ::4], g["dlr_sub"]) < 0.15
    P = dict(net.named_parameters())
    assert rel_to_max(P["block1.0.weight"].grad, g["g_block1_w"]) < 2e-2
    assert rel_to_max(P["block8.1.weight"].grad[:, ::8], g["g_block8_w_sub"]) < 2e-2
    assert rel_to_max(P["block8.1.bias"].grad, g["g_block8_b"]) < 2e-2
    top = max(v for v in gn.values() if v is not None)
    bad = [(k, float(P[k].grad.norm()), v) for k, v in gn.items()
           if v is not None and not abs(float(P[k].grad.norm()) - v) <= 2e-2 * v + 1e-4 * top]
    assert not bad, bad[:10]
    net2 = fill_module_(mod(STN=True, mask=True)).cuda().eval()
    with torch.no_grad():
        assert rel_to_max(net2(lr4.cuda()), np.load(os.path.join(golden_dir, "%s_mask_eval.npz" % arch))["sr"]) < 1e-3


def test_feature_enhancer_block_golden(golden_dir, prec_mode):
    """one FeatureEnhancer block in isolation (packed QKV projection, fused attention, std-LayerNorm with
    deferred residual gradients, FFN, 128 -> 64 linear) against the reference block: tokens, d/d feature map, every
    parameter-gradient norm.  Product layout is channel-last [B, 1024, 64]; the reference's is [B, 64, 1024]."""
    from fudanocr_amd.model.tbsrn import FeatureEnhancer
    from fudanocr_amd.utils.weight_fill import fill_module_
    g = np.load(os.path.join(golden_dir, "feature_enhancer_block.npz"))
    meta = json.load(open(os.path.join(golden_dir, "feature_enhancer_block.json")))
    fe = fill_module_(FeatureEnhancer())
    assert [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in fe.state_dict().items()] == meta["schema"]
    fe = fe.cuda().train()
    eval_dropout(fe)
    gen = torch.Generator().manual_seed(4242)
    feat = (torch.randint(0, 1 << 16, (4, 64, 1024), generator=gen).float() / float(1 << 16) - 0.5) * 2.0
    go = None
    x = feat.transpose(1, 2).contiguous().cuda().requires_grad_(True)
    out = fe(x)
    go = torch.randint(0, 1 << 16, (4, 64, 1024), generator=gen).float() / float(1 << 16) - 0.5
    out.backward(go.transpose(1, 2).contiguous().cuda())
    from fudanocr_amd import kernels as K
    K.check_deferred()
    assert rel_to_max(out.transpose(1, 2)[:, ::2, ::4], g["out_sub"]) < 1e-3
    assert rel_to_max(x.grad.transpose(1, 2)[:, ::4, ::8], g["dfeat_sub"]) < 2e-2
    P = dict(fe.named_parameters())
    # (the key-projection bias has a mathematically zero gradient -- softmax is invariant to a per-query constant --
    # so both sides hold rounding noise there: absolute floor relative to the largest norm, as in the model tests)
    top = max(v for v in meta["gradnorms"].values() if v is not None)
    bad = [(k, float(P[k].grad.norm()), v) for k, v in meta["gradnorms"].items()
           if v is not None and not abs(float(P[k].grad.norm()) - v) <= 2e-2 * v + 1e-4 * top]
    assert not bad, bad


def test_crnn_trainable_weight_gradients():
    """the recognizer with requires_grad left on (SURVEY 3.3 'switch to train them'): every parameter gradient of
    CTC(CRNN(gray)) -- convolutions, eval-mode BatchNorm affine, both BiLSTM layers (W_ih, W_hh, b_ih, b_hh of both
    directions), the two embeddings -- against torch autograd on the functional oracle with the same weights"""
    from oracle import sr_oracle as O
    from fudanocr_amd import kernels as K
    from fudanocr_amd.loss.ctc_focus_loss import CTCFocusLoss
    from fudanocr_amd.model.crnn import crnn
    from fudanocr_amd.utils.weight_fill import fill_dict_, fill_module_
    rec = fill_module_(crnn.CRNN(32, 1, 37, 256)).cuda().eval()            # eval: BatchNorm uses its running statistics
    P = fill_dict_(O.make_params(O.schema_crnn(), requires_grad=False))
    for k, v in P.items():
        if not O.is_buffer(k):
            v.requires_grad_(True)
    img = torch.rand(4, 3, 32, 128, generator=torch.Generator().manual_seed(78))
    labels = ["ab1", "hello", "zz", "0123456"]
    tg, tl = O.encode_labels(labels)
    ref_loss = O.ctc_from_logits(O.crnn_forward(P, O.parse_crnn_data(img), training=False), tg, tl)
    ref_loss.backward()
    crit = CTCFocusLoss(rec)
    logits = rec(K.bicubic_gray(img.cuda(), 100))
    enc = crit.encode(labels, logits.device)
    loss = K.ctc_loss(logits, enc[0], enc[1])
    assert abs(loss.item() - ref_loss.item()) < 1e-3 * abs(ref_loss.item())
    loss.backward()
    bad = []
    for k, p_ in rec.named_parameters():
        ref = P[k].grad
        assert p_.grad is not None, k
        err = rel_to_max(p_.grad, ref)
        if not err < 2e-2:
            bad.append((k, err))
    assert not bad, bad
    assert sum(1 for k, _ in rec.named_parameters() if ".rnn.weight_hh" in k) == 4


@pytest.mark.parametrize("arch", ["tbsrn", "tsrn"])
def test_e2e_ctc_golden(arch, golden_dir, prec_mode):
    skip_dup(arch, prec_mode)
    g = np.load(os.path.join(golden_dir, "%s_e2e_ctc.npz" % arch))
    gn = json.load(open(os.path.join(golden_dir, "%s_e2e_ctc_gradnorms.json" % arch)))
    net, rec, crit = build(arch)
    net.train()
    eval_dropout(net)
    lr, hr, labels = make_batch(4, 1234)
    sr = net(lr.cuda())
    loss, mse, _, ctc = crit(sr, hr.cuda(), labels)
    (loss * 100).backward()
    assert rel_to_max(sr, g["sr"]) < 1e-3
    assert abs(mse.item() - float(g["mse"])) < 1e-3 * float(g["mse"])
    assert abs(ctc.item() - float(g["ctc"])) < 1e-3 * float(g["ctc"])
    P = dict(net.named_parameters())
    assert rel_to_max(P["block1.0.weight"].grad, g["g_block1_w"]) < 2e-2
    top = max(v for v in gn.values() if v is not None)
    bad = [(k, float(P[k].grad.norm()), v) for k, v in gn.items()
           if v is not None and not abs(float(P[k].grad.norm()) - v) <= 2e-2 * v + 1e-4 * top]
    assert not bad, bad[:10]


@pytest.mark.parametrize("arch", ["tbsrn", "tsrn"])
def test_traj3_golden(arch, golden_dir, prec_mode):
    skip_dup(arch, prec_mode)
    """3 optimisation steps through the engine (flat buffers, fused clip+Adam) vs the reference
    models stepped with torch's own clip_grad_norm_/Adam (fixture F8)."""
    ref = json.load(open(os.path.join(golden_dir, "%s_traj3.json" % arch)))
    from fudanocr_amd.engine import TrainStep
    net, rec, crit = build(arch)
    step = TrainStep(net, crit, dropout=False)
    for s in range(3):
        lr, hr, labels = make_batch(4, 1234 + s)
        out = step(lr.cuda(), hr.cuda(), labels)
        assert abs(out["loss"].item() - ref["loss"][s]) <= 1e-3 * abs(ref["loss"][s]), (s, out["loss"].item())
        assert abs(step.opt.grad_norm().item() - ref["grad_norm"][s]) <= 2e-2 * ref["grad_norm"][s]
    sd = net.state_dict()
    bad = []
    for k, v in ref["param_abs_sum"].items():
        got = float(sd[k].double().abs().sum())
        tol = 1e-3 * v + 1e-6
        # a conv/linear bias that feeds a train-mode BatchNorm has a mathematically zero gradient: Adam turns its
        # rounding noise into +-lr steps, so two correct fp32 implementations differ by up to steps*lr per element
        if k.endswith(".bias") and (".conv1." in k or ".conv2." in k or k.startswith(("block7.0", "stn_head.stn_convnet",
                                                                                  "stn_head.stn_fc1.0"))):
            tol += 3 * 1e-4 * sd[k].numel()
        # mode 3 (single-bf16 data gradients): Adam's normalised steps turn the bf16 rounding of small gradients into
        # +-lr differences of the STN head's weights; the B = 4 batch statistics downstream of them move by ~1e-3
        if prec_mode == 3 and "running_" in k:
            tol += 2e-3 * v
        if not abs(got - v) <= tol:
            bad.append((k, got, v))
    assert not bad, bad[:10]


@pytest.mark.parametrize("batch", [8])
def test_step_vs_oracle_fresh_batch(batch, prec_mode):
    """One full step (TBSRN + CRNN + CTC) on a fresh seeded batch vs the CPU oracle."""
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    net, rec, crit = build()
    step = TrainStep(net, crit, dropout=False)
    lr, hr, labels = make_batch(batch, 99)
    out = step(lr.cuda(), hr.cuda(), labels)
    P = O.make_params(O.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    C = O.make_params(O.schema_crnn(), requires_grad=False)
    fill_dict_(C)
    opt = O.AdamState([v for v in P.values() if v.requires_grad])
    tgt, tlen = O.encode_labels(labels)
    r = O.train_step(P, opt, "tbsrn", lr, hr, C, tgt, tlen)
    assert rel_to_max(out["sr"], r["sr"]) < 1e-3
    assert abs(out["loss"].item() - r["loss"]) < 1e-3 * abs(r["loss"])
    assert abs(out["ctc"].item() - r["ctc"]) < 1e-3 * abs(r["ctc"])
    assert abs(step.opt.grad_norm().item() - r["grad_norm"]) < 2e-2 * r["grad_norm"]


# allowance of test_gradients_elementwise_b32_fp64 per arithmetic mode (measured margins: profiles/r05_test_margins.txt)
GRAD_ALLOWANCE_B32 = {2: 1e-3, 3: 9e-3}

SMOOTH_PARAMS = ("conv1.weight", "conv2.weight", ".pff.w_1.weight", ".pff.w_2.weight", ".linears.0.weight",
                 ".linears.1.weight", ".linears.2.weight", ".linears.3.weight", ".feature_enhancer.linear.weight",
                 ".mul_layernorm1.a_2", ".mul_layernorm1.b_2", ".mul_layernorm3.a_2", ".mul_layernorm3.b_2",
                 ".pff.w_1.bias", ".linears.3.bias", "block7.0.weight", "block8.0.conv.weight", "block8.1.weight")


@pytest.mark.parametrize("with_ctc", [False, True], ids=["mse", "e2e-ctc"])
def test_train_gradients_elementwise_vs_oracle(with_ctc, prec_mode):
    """Element-wise parameter gradients (not only their norms) of the B = 4 golden batch against the pinned CPU oracle,
    on every smooth parameter of the SR trunk: SRB convolutions, Q/K/V/O, FFN, LayerNorm a_2 / b_2, block7 / block8.
    (The STN head sits behind the bilinear-cell decisions of the TPS sampler and stays norm-gated, see
    test_train_mse_golden.)  <= 1e-2 of each gradient's max, in every precision mode incl. the bench default (3)."""
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    net, rec, crit = build("tbsrn")
    net.train()
    eval_dropout(net)
    lr, hr, labels = make_batch(4, 1234)
    if with_ctc:
        loss = crit(net(lr.cuda()), hr.cuda(), labels)[0]
    else:
        from fudanocr_amd import kernels as K
        loss = K.mse_loss(net(lr.cuda()), hr.cuda())
    (loss * 100).backward()
    P = O.make_params(O.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    C = tgt = tlen = None
    if with_ctc:
        C = O.make_params(O.schema_crnn(), requires_grad=False)
        fill_dict_(C)
        tgt, tlen = O.encode_labels(labels)
    oloss, _, _, _ = O.step_loss(P, "tbsrn", lr, hr, C, tgt, tlen, True, 0.0, 5, True)
    (oloss * 100).backward()
    assert abs(loss.item() - oloss.item()) < 1e-3 * abs(oloss.item())
    # the same oracle in float64 = the arbitration: the fp32 oracle is itself one fp32 implementation of a B = 4
    # BatchNorm chain that amplifies rounding, so "HIP vs fp32 oracle" mixes two errors.  Per parameter:
    #   err(HIP, fp64) <= 2 x err(fp32 oracle, fp64) + allowance(mode)
    # Measured (gpurun_out/test_margins.txt): in modes 1 / 2 the worst HIP gradient is EXACTLY as far from the fp64 truth
    # as the fp32 oracle is (block6 pff.w_1.weight: 8.12e-3 vs 8.10e-3; pff.w_1.bias 6.6e-3 vs 8.3e-3) -- the 6-8e-3 that
    # does not move between the arithmetic modes is fp32-vs-fp32 noise of the B = 4 BatchNorm chain, not kernel error;
    # allowance 2e-3 (a floor for parameters on which the fp32 oracle happens to be exact).  Mode 3 rounds the operands
    # of the 3x3 data-gradient convolutions to bf16 (up to ten of them in a chain above block2): 4e-3 .. 7.4e-3 on
    # block2's parameters where the fp32 oracle is at 9e-4 -- allowance 7e-3, the price of that mode, stated here.
    P64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in P.items()}
    C64 = None if C is None else {k: (v.double() if v.is_floating_point() else v) for k, v in C.items()}
    pe0 = O.positional_encoding_2d
    O.positional_encoding_2d = lambda *a: pe0(*a).double()
    try:
        l64, _, _, _ = O.step_loss(P64, "tbsrn", lr.double(), hr.double(), C64, tgt, tlen, True, 0.0, 5, True)
        (l64 * 100).backward()
    finally:
        O.positional_encoding_2d = pe0
    assert abs(loss.item() - l64.item()) < 1e-3 * abs(l64.item())
    checked, bad, errs = 0, [], []
    bad64, pairs = [], []
    for name, p in net.named_parameters():
        if not (name.startswith("block") and ".gru" not in name and any(name.endswith(sfx) for sfx in SMOOTH_PARAMS)):
            continue
        ref = P[name].grad
        if ref is None or p.grad is None:
            continue
        checked += 1
        e = rel_to_max(p.grad, ref)
        errs.append((name, e))
        # measured worst case: 6e-3 (weights, modes 1-2), 8e-3 (mode 3); pff.w_1.bias 8.3e-3 in EVERY mode (9.7e-3 in mode
        # 3): its gradient is the column sum of the relu-masked FFN gradient, and the mask of pre-activations within
        # rounding of zero differs between any two fp32 implementations -> that one parameter kind gets 1.5e-2
        if not e <= (1.5e-2 if name.endswith(".pff.w_1.bias") else 1e-2):
            bad.append((name, e))
        truth = P64[name].grad
        e_hip, e_o32 = rel_to_max(p.grad, truth), rel_to_max(ref, truth)
        pairs.append((name, e_hip, e_o32))
        if not e_hip <= 2.0 * e_o32 + (7e-3 if prec_mode == 3 else 2e-3):
            bad64.append((name, e_hip, e_o32))
    worst = sorted(pairs, key=lambda t: -t[1])[:3]
    _note("gradients_elementwise ctc=%s mode %d: %d parameters, worst vs fp32 oracle %s; vs fp64 oracle (HIP, fp32 oracle) %s"
          % (with_ctc, prec_mode, checked, sorted(errs, key=lambda t: -t[1])[:3],
             [(n, "%.2e" % a, "%.2e" % b) for n, a, b in worst]))
    assert checked >= 60, checked
    assert not bad, bad[:10]
    assert not bad64, ("HIP gradient further from the fp64 oracle than 2 x the fp32 oracle + allowance", bad64[:10])


def test_traj_fixed_batch_vs_oracle(prec_mode):
    """Five optimisation steps on ONE fixed batch, engine vs the oracle's train_step: with the batch fixed the loss
    sequence depends on the UPDATES only (test_traj3_golden's batches change every step, so its losses are batch-driven);
    a wrong gradient direction shows up as a diverging loss curve and in the parameter displacement."""
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    net, rec, crit = build("tbsrn")
    p0 = {k: v.detach().clone() for k, v in net.named_parameters()}
    step = TrainStep(net, crit, dropout=False)
    lr, hr, labels = make_batch(4, 1234)
    P = O.make_params(O.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    q0 = {k: v.detach().clone() for k, v in P.items()}
    C = O.make_params(O.schema_crnn(), requires_grad=False)
    fill_dict_(C)
    opt = O.AdamState([v for v in P.values() if v.requires_grad])
    tgt, tlen = O.encode_labels(labels)
    got, want = [], []
    for _ in range(5):
        got.append(step(lr.cuda(), hr.cuda(), labels)["loss"].item())
        want.append(O.train_step(P, opt, "tbsrn", lr, hr, C, tgt, tlen)["loss"])
    # the five updates lower the loss by ~1.4 % on this batch (13.514 -> 13.324): 14x the tolerance of the comparison
    assert want[0] - want[-1] > 1e-2 * want[0], want
    for a, b in zip(got, want):
        assert abs(a - b) <= 1e-3 * abs(b), (got, want)
    # displacement of the smooth parameters after the five updates: direction and size
    worst = 1.0
    for name, p in net.named_parameters():
        if name.startswith("block") and ".gru" not in name and any(name.endswith(sfx) for sfx in SMOOTH_PARAMS[:9]):
            d_got = (p.detach().cpu() - p0[name].cpu()).flatten().double()
            d_ref = (P[name].detach() - q0[name]).flatten().double()
            cos = float(torch.dot(d_got, d_ref) / (d_got.norm() * d_ref.norm() + 1e-30))
            worst = min(worst, cos)
    # Adam normalises every element's step to ~lr, so elements whose gradient is below the arithmetic's noise floor move
    # in a random direction in ANY two correct implementations: fp32 vs fp64 oracle gives 0.997 here
    _note("traj_fixed_batch mode %d: worst displacement cosine %.4f, losses %s" % (prec_mode, worst, got))
    assert worst > 0.9, worst


class _oracle_threads:
    """the CPU oracle at BASELINE sizes: 32 threads (beyond that these 16x64-pixel ops only slow down on the 256-thread host,
    bench.py cpu_baseline measures both), restored afterwards"""

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def _oracle_fp64(O, P, C, lr, hr, tgt, tlen):
    """the same oracle in float64 (the arbitration leg, see test_train_gradients_elementwise_vs_oracle)"""
    P64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v) for k, v in P.items()}
    C64 = None if C is None else {k: (v.double() if v.is_floating_point() else v) for k, v in C.items()}
    pe0 = O.positional_encoding_2d
    O.positional_encoding_2d = lambda *a: pe0(*a).double()
    try:
        l64, _, _, _ = O.step_loss(P64, "tbsrn", lr.double(), hr.double(), C64, tgt, tlen, True, 0.0, 5, True)
        (l64 * 100).backward()
    finally:
        O.positional_encoding_2d = pe0
    return P64, l64


@pytest.mark.parametrize("cfg", ["c3", "c2"])
def test_full_size_step_vs_oracle(cfg):
    """BASELINE configs[2] (TBSRN + frozen CRNN-CTC, per-GPU batch 128) and configs[1] (TBSRN, MSE only, batch 64) at
    their FULL sizes against the CPU oracle's step (interfaces/super_resolution.py:69-84 restated in oracle.sr_oracle),
    in the bench's arithmetic mode (3), dropout slots in eval: SR pixels <= 1e-3 of max, MSE / CTC / loss <= 1e-3,
    pre-clip gradient norm <= 2e-2, and every smooth trunk parameter's gradient element-wise (<= 1e-2 of its max: at
    these batch sizes the BatchNorm chain no longer amplifies fp32 rounding, so this gate measures the kernels)."""
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.loss.ctc_focus_loss import CTCFocusLoss
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    b = 128 if cfg == "c3" else 64
    old = _lib.get_precision()
    _lib.set_precision(3)
    try:
        net, rec, crit = build()
        if cfg == "c2":
            crit = CTCFocusLoss(None)
        step = TrainStep(net, crit, dropout=False)
        lr, hr, labels = make_batch(b, 2025)
        out = step(lr.cuda(), hr.cuda(), labels) if cfg == "c3" else step(lr.cuda(), hr.cuda())
        torch.cuda.synchronize()
        P = O.make_params(O.schema_sr("tbsrn"))
        fill_dict_({k: v.data for k, v in P.items()})
        C = tgt = tlen = None
        if cfg == "c3":
            C = O.make_params(O.schema_crnn(), requires_grad=False)
            fill_dict_(C)
            tgt, tlen = O.encode_labels(labels)
        with _oracle_threads():
            loss, mse, ctc, sr = O.step_loss(P, "tbsrn", lr, hr, C, tgt, tlen, True, 0.0, 5, True)
            (loss * 100).backward()
        gn = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in P.values() if v.requires_grad and v.grad is not None)))
        e_sr = rel_to_max(out["sr"], sr.detach())
        e_loss = abs(out["loss"].item() - loss.item()) / abs(loss.item())
        e_mse = abs(out["mse"].item() - mse.item()) / abs(mse.item())
        e_ctc = abs(out["ctc"].item() - ctc.item()) / abs(ctc.item()) if cfg == "c3" else 0.0
        e_gn = abs(step.opt.grad_norm().item() - gn) / gn
        errs = []
        for name, p in net.named_parameters():
            if name.startswith("block") and ".gru" not in name and any(name.endswith(sfx) for sfx in SMOOTH_PARAMS):
                if P[name].grad is not None and p.grad is not None:
                    errs.append((name, rel_to_max(p.grad, P[name].grad)))
        worst = sorted(errs, key=lambda t: -t[1])[:3]
        _note("full_size_step_vs_oracle %s (B = %d, mode 3): sr %.2e loss %.2e mse %.2e ctc %.2e grad-norm %.2e; %d gradients "
              "element-wise, worst %s" % (cfg, b, e_sr, e_loss, e_mse, e_ctc, e_gn, len(errs),
                                          [(n, "%.2e" % e) for n, e in worst]))
        assert e_sr < 1e-3 and e_loss < 1e-3 and e_mse < 1e-3 and e_ctc < 1e-3, (e_sr, e_loss, e_mse, e_ctc)
        assert e_gn < 2e-2, e_gn
        assert len(errs) >= 60 and worst[0][1] < 1e-2, worst
    finally:
        _lib.set_precision(old)


def test_full_size_step_vs_oracle_mask():
    """The reference's `--mask` (main.py:31: a fourth, binarised-luma input / output channel, dataset.py:146-151) on the c3
    step at B = 64 against the CPU oracle: both 9 x 9 layers run on their four-channel kernels (conv9x9_cin4.hip forward and
    output-layer data gradient; conv9x9_out.hip as two launches of two channels, forward and weight gradient).  SR pixels,
    losses, pre-clip gradient norm, and the gradients of the two 9 x 9 layers element-wise."""
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.smoke import build_models
    from fudanocr_amd.utils.synth import with_mask
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    b = 64
    old = _lib.get_precision()
    _lib.set_precision(3)
    try:
        net, rec, crit = build_models(torch.device("cuda:0"), "tbsrn", mask=True)
        step = TrainStep(net, crit, dropout=False)
        lr, hr, labels = make_batch(b, 2027)
        lr, hr = with_mask(lr), with_mask(hr)
        out = step(lr.cuda(), hr.cuda(), labels)
        torch.cuda.synchronize()
        P = O.make_params(O.schema_sr("tbsrn", in_planes=4))
        fill_dict_({k: v.data for k, v in P.items()})
        C = O.make_params(O.schema_crnn(), requires_grad=False)
        fill_dict_(C)
        tgt, tlen = O.encode_labels(labels)
        with _oracle_threads():
            loss, mse, ctc, sr = O.step_loss(P, "tbsrn", lr, hr, C, tgt, tlen, True, 0.0, 5, True)
            (loss * 100).backward()
        gn = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in P.values() if v.requires_grad and v.grad is not None)))
        e_sr = rel_to_max(out["sr"], sr.detach())
        e_loss = abs(out["loss"].item() - loss.item()) / abs(loss.item())
        e_ctc = abs(out["ctc"].item() - ctc.item()) / abs(ctc.item())
        e_gn = abs(step.opt.grad_norm().item() - gn) / gn
        named = dict(net.named_parameters())
        nine = [n for n, p in named.items() if p.dim() == 4 and p.shape[-1] == 9]
        errs = [(n, rel_to_max(named[n].grad, P[n].grad)) for n in nine]
        _note("full_size_step_vs_oracle --mask (B = %d, mode 3): sr %.2e loss %.2e ctc %.2e grad-norm %.2e; 9x9 weight gradients %s"
              % (b, e_sr, e_loss, e_ctc, e_gn, [(n, tuple(named[n].shape), "%.2e" % e) for n, e in errs]))
        assert out["sr"].shape[1] == 4 and len(nine) == 2, nine
        assert e_sr < 1e-3 and e_loss < 1e-3 and e_ctc < 1e-3, (e_sr, e_loss, e_ctc)
        assert e_gn < 2e-2, e_gn
        assert max(e for _, e in errs) < 1e-2, errs
    finally:
        _lib.set_precision(old)


def test_full_size_step_vs_oracle_tsrn():
    """The TSRN variant of configs[2] (`bench.py --config c1`: TSRN + frozen CRNN-CTC, per-GPU batch 128) at its FULL size
    against the CPU oracle's step, in the bench's arithmetic mode (3): SR pixels, losses, pre-clip gradient norm, and the
    gradients of the trunk convolutions AND of every GruBlock tensor element-wise -- the recurrent scans (16-sequence
    loader / compute waves), W_ih / 1x1 weight gradients on the streaming kernel's K = 64 mode and W_hh from the
    cross-product pass all sit on this path (model/tsrn.py:89-98,128-145 of the reference)."""
    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    b = 128
    old = _lib.get_precision()
    _lib.set_precision(3)
    try:
        net, rec, crit = build("tsrn")
        step = TrainStep(net, crit, dropout=False)
        lr, hr, labels = make_batch(b, 2026)
        out = step(lr.cuda(), hr.cuda(), labels)
        torch.cuda.synchronize()
        P = O.make_params(O.schema_sr("tsrn"))
        fill_dict_({k: v.data for k, v in P.items()})
        C = O.make_params(O.schema_crnn(), requires_grad=False)
        fill_dict_(C)
        tgt, tlen = O.encode_labels(labels)
        with _oracle_threads():
            loss, mse, ctc, sr = O.step_loss(P, "tsrn", lr, hr, C, tgt, tlen, True, 0.0, 5, True)
            (loss * 100).backward()
        gn = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for v in P.values() if v.requires_grad and v.grad is not None)))
        e_sr = rel_to_max(out["sr"], sr.detach())
        e_loss = abs(out["loss"].item() - loss.item()) / abs(loss.item())
        e_ctc = abs(out["ctc"].item() - ctc.item()) / abs(ctc.item())
        e_gn = abs(step.opt.grad_norm().item() - gn) / gn
        errs = []
        for name, p in net.named_parameters():
            if not name.startswith("block") or P[name].grad is None or p.grad is None:
                continue
            if ".gru" in name or name.endswith(("conv1.weight", "conv2.weight")):
                errs.append((name, rel_to_max(p.grad, P[name].grad)))
        gru = [e for n, e in errs if ".gru" in n]
        worst = sorted(errs, key=lambda t: -t[1])[:4]
        _note("full_size_step_vs_oracle c1 (TSRN, B = %d, mode 3): sr %.2e loss %.2e ctc %.2e grad-norm %.2e; %d gradients "
              "element-wise (%d GruBlock tensors), worst %s" % (b, e_sr, e_loss, e_ctc, e_gn, len(errs), len(gru),
                                                               [(n, "%.2e" % e) for n, e in worst]))
        assert e_sr < 1e-3 and e_loss < 1e-3 and e_ctc < 1e-3, (e_sr, e_loss, e_ctc)
        assert e_gn < 2e-2, e_gn
        assert len(gru) >= 80 and worst[0][1] < 1.5e-2, worst          # measured 9.5e-3 (block2, behind ten bf16 data-gradient layers)
    finally:
        _lib.set_precision(old)


@pytest.mark.parametrize("mode", [2, 3], ids=["fastgrad", "dgrad16"])
def test_gradients_elementwise_b32_fp64(mode):
    """The element-wise gradient gate at a batch where the BatchNorm chain's fp32-vs-fp32 noise (6-8e-3 at B = 4, see
    test_train_gradients_elementwise_vs_oracle) is gone: B = 32, TBSRN + CRNN-CTC, fp64 oracle as the truth.  Per smooth
    parameter err(HIP, fp64) <= 2 x err(fp32 oracle, fp64) + allowance; the allowance is what the ARITHMETIC MODE costs
    (mode 2: split products except the attention's gradient accumulations; mode 3: + single-bf16 data-gradient
    convolutions and dP), measured (profiles/r05_test_margins.txt) and stated here.  Mode 2: the worst HIP gradient is 1.9e-3
    from the fp64 truth where the fp32 CPU oracle is at 2.6e-3 (pff.w_1.bias of block2) -- never further than the fp32
    oracle + 4e-4: allowance 1e-3.  Mode 3: 7.5e-3 on block2's q / k projection weights (fp32 oracle 3e-4): ten bf16
    data-gradient convolutions and five bf16 dP products lie between the loss and that block -- this IS the price of the
    mode at a batch size where BatchNorm noise no longer hides it; allowance 9e-3."""
    from fudanocr_amd import _lib
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    old = _lib.get_precision()
    _lib.set_precision(mode)
    try:
        net, rec, crit = build("tbsrn")
        net.train()
        eval_dropout(net)
        lr, hr, labels = make_batch(32, 4321)
        loss = crit(net(lr.cuda()), hr.cuda(), labels)[0]
        (loss * 100).backward()
        torch.cuda.synchronize()
        P = O.make_params(O.schema_sr("tbsrn"))
        fill_dict_({k: v.data for k, v in P.items()})
        C = O.make_params(O.schema_crnn(), requires_grad=False)
        fill_dict_(C)
        tgt, tlen = O.encode_labels(labels)
        with _oracle_threads():
            oloss, _, _, _ = O.step_loss(P, "tbsrn", lr, hr, C, tgt, tlen, True, 0.0, 5, True)
            (oloss * 100).backward()
            P64, l64 = _oracle_fp64(O, P, C, lr, hr, tgt, tlen)
        assert abs(loss.item() - l64.item()) < 1e-3 * abs(l64.item())
        allowance = GRAD_ALLOWANCE_B32[mode]
        pairs, bad = [], []
        for name, p in net.named_parameters():
            if not (name.startswith("block") and ".gru" not in name and any(name.endswith(sfx) for sfx in SMOOTH_PARAMS)):
                continue
            if P[name].grad is None or p.grad is None:
                continue
            truth = P64[name].grad
            e_hip, e_o32 = rel_to_max(p.grad, truth), rel_to_max(P[name].grad, truth)
            pairs.append((name, e_hip, e_o32))
            if not e_hip <= 2.0 * e_o32 + allowance:
                bad.append((name, e_hip, e_o32))
        worst = sorted(pairs, key=lambda t: -t[1])[:4]
        _note("gradients_elementwise B = 32 mode %d: %d parameters, worst vs fp64 oracle (HIP, fp32 oracle) %s; worst fp32 oracle "
              "%.2e" % (mode, len(pairs), [(n, "%.2e" % a, "%.2e" % b_) for n, a, b_ in worst], max(t[2] for t in pairs)))
        assert len(pairs) >= 60, len(pairs)
        assert not bad, bad[:10]
    finally:
        _lib.set_precision(old)


def test_eval_batchnorm_follows_training(golden_dir):
    """the eval-mode BatchNorm caches 1/sqrt(running_var + eps) per tensor; the train kernels rewrite
    running_var through raw pointers (no autograd version bump), so a train-mode forward must drop the cached value.
    eval -> train steps -> eval must equal a fresh module loaded with the same state_dict (TextSR.eval after
    TextSR.train, interfaces/super_resolution.py:161-171)."""
    from fudanocr_amd.engine import TrainStep
    net, rec, crit = build()
    lr, hr, labels = make_batch(4, 1234)
    net.eval()
    with torch.no_grad():
        sr0 = net(lr.cuda()).clone()                       # fills the cache
    step = TrainStep(net, crit, dropout=False)
    for i in range(3):
        l2, h2, lab2 = make_batch(8, 50 + i)
        step(l2.cuda(), h2.cuda(), lab2)
    net.eval()
    with torch.no_grad():
        sr1 = net(lr.cuda()).clone()
    fresh, _, _ = build()
    fresh.load_state_dict({k: v.clone() for k, v in net.state_dict().items()})
    fresh.eval()
    with torch.no_grad():
        sr2 = fresh(lr.cuda())
    assert not torch.equal(sr0, sr1)                       # training changed the running statistics
    assert torch.equal(sr1, sr2), (sr1 - sr2).abs().max().item()


@pytest.mark.parametrize("cfg", ["c3", "c2"])
def test_full_size_properties(cfg):
    """BASELINE full sizes (c3: TBSRN + CRNN-CTC at per-GPU batch 128; c2: TBSRN, MSE only, batch 64): size-independent
    checks -- finite loss, output in (-1,1), the loss decreases over a few steps on a fixed batch -- and determinism:
    two engines stepped on the same inputs with dropout off agree (forward bit-identical: no atomics; parameters after
    the step up to the atomics noise of the backward reductions)."""
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.loss.ctc_focus_loss import CTCFocusLoss
    b = 128 if cfg == "c3" else 64
    lr, hr, labels = make_batch(b, 7)
    lr, hr = lr.cuda(), hr.cuda()

    def engine(dropout):
        net, rec, crit = build()
        if cfg == "c2":
            crit = CTCFocusLoss(None)
        return net, TrainStep(net, crit, dropout=dropout), crit
    net, step, crit = engine(True)
    enc = crit.encode(labels, lr.device) if cfg == "c3" else None
    losses = []
    for _ in range(6):
        out = step(lr, hr, encoded=enc) if cfg == "c3" else step(lr, hr)
        losses.append(out["loss"].item())
    assert all(np.isfinite(losses)), losses
    assert out["sr"].abs().max().item() < 1.0
    assert losses[-1] < losses[0], losses
    outs = []
    for _ in range(2):
        net_d, step_d, crit_d = engine(False)
        o = step_d(lr, hr, encoded=enc) if cfg == "c3" else step_d(lr, hr)
        outs.append((o["sr"].clone(), o["loss"].item(), step_d.flat.flat_param.clone()))
    assert torch.equal(outs[0][0], outs[1][0])                  # SR pixels: bit-identical (no atomics in the forward)
    assert abs(outs[0][1] - outs[1][1]) <= 1e-6 * abs(outs[0][1])   # the MSE scalar is summed with block-level atomics
    # one Adam step moves every element by <= lr = 1e-4; gradient-order noise can flip the step of near-zero gradients
    assert (outs[0][2] - outs[1][2]).abs().max().item() <= 2.5e-4
    assert (outs[0][2] - outs[1][2]).abs().mean().item() <= 2e-6


def test_harness_train_eval_checkpoint(tmp_path, monkeypatch):
    """TextSR harness (reference main.py -> interfaces): 3 synthetic training iterations, eval (PSNR/SSIM/
    accuracy), checkpoint in the reference schema, reload into a fresh model."""
    import os
    import yaml
    from fudanocr_amd import main as M
    from fudanocr_amd.utils.util import AttrDict
    monkeypatch.chdir(tmp_path)
    cfg = AttrDict(yaml.load(open(os.path.join(os.path.dirname(M.__file__), "config", "super_resolution.yaml")),
                             Loader=yaml.Loader))
    cfg.TRAIN.iters_per_epoch, cfg.TRAIN.displayInterval, cfg.TRAIN.saveInterval = 3, 1, 2
    args = M.parse(["--arch", "tbsrn", "--STN", "--exp_name", "t", "--batch_size", "8"])
    res = M.main(cfg, args)
    assert res["images_per_sec"] > 0
    ck = torch.load(tmp_path / "checkpoint" / "t" / "model_best.pth", weights_only=False)
    assert set(ck) == {"state_dict_G", "info", "best_history_res", "best_model_info", "param_num", "converge"}
    assert ck["info"]["arch"] == "tbsrn" and ck["param_num"] == 3210335 and len(ck["state_dict_G"]) == 346
    from fudanocr_amd.model import tbsrn
    m = tbsrn.TBSRN(STN=True)
    m.load_state_dict(ck["state_dict_G"])
    conv = ck["converge"][-1]
    assert 0 <= conv["acc"] <= 1 and conv["psnr"] > 0 and -1 <= conv["ssim"] <= 1


def test_harness_demo_and_dispatch(tmp_path, monkeypatch):
    """main.py dispatches --demo to TextSR.demo() (reference main.py:8-15, interfaces/super_resolution.py:331-420):
    every image of the demo directory is resized to 64 x 16, super-resolved and recognised from LR and from SR."""
    import os
    import yaml
    from PIL import Image
    from fudanocr_amd import main as M
    from fudanocr_amd.utils.util import AttrDict
    monkeypatch.chdir(tmp_path)
    demo = tmp_path / "demo"
    demo.mkdir()
    rng = np.random.RandomState(3)
    for i, (w, h) in enumerate([(100, 31), (64, 16), (200, 60)]):
        Image.fromarray(rng.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(demo / ("im%d.png" % i))
    cfg = AttrDict(yaml.load(open(os.path.join(os.path.dirname(M.__file__), "config", "super_resolution.yaml")),
                             Loader=yaml.Loader))
    for extra in ([], ["--mask"]):
        args = M.parse(["--arch", "tbsrn", "--STN", "--exp_name", "d", "--demo", "--demo_dir", str(demo)] + extra)
        res = M.main(cfg, args)
        assert [r[0] for r in res["results"]] == ["im0.png", "im1.png", "im2.png"]
        assert all(isinstance(a, str) and isinstance(b, str) for _, a, b in res["results"]) and res["fps"] > 0
    assert not (tmp_path / "checkpoint").exists()              # a demo run never touches the checkpoint directory


def test_eval_metrics_on_device(golden_dir):
    """row N4: PSNR / SSIM from the fused HIP pass against fixture F9 (values produced by the reference's
    utils/ssim_psnr.py) and, on other shapes / the per-image form, against the torch formula in float64"""
    from fudanocr_amd.utils import ssim_psnr
    ref = json.load(open(os.path.join(golden_dir, "metrics.json")))
    g = torch.Generator().manual_seed(11)
    ia = torch.rand(3, 3, 32, 128, generator=g)
    ib = (ia + 0.1 * torch.randn(3, 3, 32, 128, generator=g)).clamp(0, 1)
    assert abs(float(ssim_psnr.calculate_psnr(ia.cuda(), ib.cuda())) - ref["psnr"]) < 1e-4
    assert abs(float(ssim_psnr.SSIM()(ia.cuda(), ib.cuda())) - ref["ssim"]) < 1e-5
    p, s = ssim_psnr.psnr_ssim(ia.cuda(), ib.cuda())
    assert abs(float(p) - ref["psnr"]) < 1e-4 and abs(float(s) - ref["ssim"]) < 1e-5
    # 4-channel (mask) input: only the first three channels count; odd sizes; per-image means
    a = torch.rand(5, 4, 17, 37, generator=g)
    b = (a + 0.05 * torch.randn(5, 4, 17, 37, generator=g)).clamp(0, 1)
    want = ssim_psnr.SSIM(size_average=False)(a.double(), b.double())
    got = ssim_psnr.SSIM(size_average=False)(a.cuda(), b.cuda())
    assert (got.cpu().double() - want).abs().max() < 1e-5
    assert abs(float(ssim_psnr.calculate_psnr(a.cuda(), b.cuda())) - float(ssim_psnr.calculate_psnr(a.double(), b.double()))) < 1e-4
    with pytest.raises(RuntimeError):
        ssim_psnr.calculate_psnr(a.cuda(), b.cuda()[:, :2])
    vals = {float(ssim_psnr.SSIM()(ia.cuda(), ib.cuda())) for _ in range(3)}
    assert len(vals) == 1

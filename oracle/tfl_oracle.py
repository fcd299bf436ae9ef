"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Functional fp32 PyTorch-CPU restatement of the text-focus loss: the frozen transformer recognizer of
scene-text-telescope/loss/transformer.py:82-389 and the three-term criterion of loss/text_focus_loss.py:54-104 with
loss/weight_ce_loss.py:38-45.  Pure functions of a flat {state_dict key: tensor} dict in the reference's schema
(tests/golden/tfl_schema.json); shared building blocks come from oracle/sld_oracle.py (the two transformers are
siblings).  Pinned by tests/test_text_focus.py against fixture tfl_step.npz (tools/make_golden_tfl.py)."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import sld_oracle as S

ALPHABET = "-0123456789abcdefghijklmnopqrstuvwxyz"          # loss/transformer.py:8
ENGLISH = "-0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"   # text_focus_loss.py:47
LAYERS = (1, 2, 5, 3)                                       # transformer.py:333


def schema(n_class=37, max_len=5000):
    d = OrderedDict()
    d["embedding_word.lut.weight"] = torch.zeros(n_class, 512)
    d["pe.pe"] = S.positional_table(512, max_len)[None]
    e = "encoder.cnn."
    S._conv(d, e + "conv1.", 64, 1)
    S._bn(d, e + "bn1.", 64)
    S._conv(d, e + "conv2.", 128, 64)
    S._bn(d, e + "bn2.", 128)
    plan = ((1, 128, 256, 256), (2, 256, 256, 256), (3, 256, 512, 512), (4, 512, 512, 1024))
    for (li, cin, planes, cout), nb in zip(plan, LAYERS):
        S._layer(d, e + "layer%d." % li, cin, planes, nb)
        tail = "layer%d_conv." % li if li < 4 else "layer4_conv2."
        tbn = "layer%d_bn." % li if li < 4 else "layer4_conv2_bn."
        S._conv(d, e + tail, cout, planes)
        S._bn(d, e + tbn, cout)
    out = d                                                  # decoder in the reference's registration order
    S._mha(out, "decoder.mask_multihead.", 1024, 16)
    out["decoder.mul_layernorm1.a_2"], out["decoder.mul_layernorm1.b_2"] = torch.ones(1024), torch.zeros(1024)
    S._mha(out, "decoder.multihead.", 1024, 16)
    out["decoder.mul_layernorm2.a_2"], out["decoder.mul_layernorm2.b_2"] = torch.ones(1024), torch.zeros(1024)
    S._lin(out, "decoder.pff.w_1.", 2048, 1024)
    S._lin(out, "decoder.pff.w_2.", 1024, 2048)
    out["decoder.mul_layernorm3.a_2"], out["decoder.mul_layernorm3.b_2"] = torch.ones(1024), torch.zeros(1024)
    S._lin(out, "generator_word.proj.", n_class, 1024)
    return out


def make_params():
    return OrderedDict((k, v.clone()) for k, v in schema().items())          # frozen: nothing requires grad


def to_gray(x):
    """text_focus_loss.py:17-22"""
    return 0.299 * x[:, 0:1] + 0.587 * x[:, 1:2] + 0.114 * x[:, 2:3]


def label_encoder(labels):
    """text_focus_loss.py:62-81 on labels already str_filt'ed + '-'"""
    a2n = {c: i for i, c in enumerate(ENGLISH)}
    length = torch.tensor([len(s) for s in labels], dtype=torch.long)
    text_input = torch.zeros(len(labels), int(length.max()), dtype=torch.long)
    for i, s in enumerate(labels):
        for j in range(len(s) - 1):
            text_input[i, j + 1] = a2n[s[j]]
    text_gt = torch.tensor([a2n[c] for s in labels for c in s], dtype=torch.long)
    return length, text_input, text_gt


def encoder(P, x):
    """loss/transformer.py:133-163: TWO max-pools (after conv1 and in front of layer1): 32x128 -> 8x32; eval-mode BN"""
    e = "encoder.cnn."
    x = F.max_pool2d(F.relu(S._bnf(P, e + "bn1.", S._cv(P, e + "conv1.", x), False)), 2, 2)
    x = F.relu(S._bnf(P, e + "bn2.", S._cv(P, e + "conv2.", x), False))
    x = F.max_pool2d(x, 2, 2)
    for li, nb in zip((1, 2, 3, 4), LAYERS):
        for i in range(nb):
            x = S.basic_block(P, "%slayer%d.%d." % (e, li, i), x, False)
        tail = "layer%d_conv." % li if li < 4 else "layer4_conv2."
        tbn = "layer%d_bn." % li if li < 4 else "layer4_conv2_bn."
        x = F.relu(S._bnf(P, e + tbn, S._cv(P, e + tail, x), False))
    return x


def recognizer(P, gray, text_length, text_input):
    """loss/transformer.py:354-389 (test=False branch), eval mode (build_up_transformer: transformer.eval())"""
    conv = encoder(P, gray)
    emb = F.embedding(text_input, P["embedding_word.lut.weight"]) * math.sqrt(512)
    pos = P["pe.pe"][:, :emb.shape[1]].expand(emb.shape[0], -1, -1)
    x = torch.cat([emb, pos], 2)
    L = x.shape[1]
    mask = torch.tril(torch.ones(1, L, L, dtype=torch.bool))
    r = S.layernorm(x + S.mha(P, "decoder.mask_multihead.", x, x, x, mask, 0.0, h=16)[0],
                    P["decoder.mul_layernorm1.a_2"], P["decoder.mul_layernorm1.b_2"])
    b, c, hh, ww = conv.shape
    mem = conv.view(b, c, hh * ww).permute(0, 2, 1).contiguous()
    align, amap = S.mha(P, "decoder.multihead.", r, mem, mem, None, 0.0, h=16)
    r = S.layernorm(r + align, P["decoder.mul_layernorm2.a_2"], P["decoder.mul_layernorm2.b_2"])
    ff = F.linear(F.relu(F.linear(r, P["decoder.pff.w_1.weight"], P["decoder.pff.w_1.bias"])),
                  P["decoder.pff.w_2.weight"], P["decoder.pff.w_2.bias"])
    r = S.layernorm(r + ff, P["decoder.mul_layernorm3.a_2"], P["decoder.mul_layernorm3.b_2"])
    logits = F.linear(r, P["generator_word.proj.weight"], P["generator_word.proj.bias"])
    pred = torch.cat([logits[i, :int(n)] for i, n in enumerate(text_length)], 0)
    return pred, amap, conv


def weight_cross_entropy(pred, gt, table):
    """loss/weight_ce_loss.py:38-45: -mean log( w[gt][gt] e^pred[gt] / sum_c w[gt][c] e^pred[c] )"""
    w = table[gt]
    pe = w * torch.exp(pred)
    return -(torch.log(pe.gather(1, gt[:, None])[:, 0] / pe.sum(1))).sum() / gt.shape[0]


def text_focus_loss(P, sr, hr, labels_filtered, table):
    """text_focus_loss.py:84-99: mse + 10 L1(attention maps) + 0.0005 weighted CE; labels already filtered + '-'"""
    mse = F.mse_loss(sr, hr)
    length, text_input, text_gt = label_encoder(labels_filtered)
    with torch.no_grad():
        _, map_gt, _ = recognizer(P, to_gray(hr), length, text_input)
    pred, map_pred, conv = recognizer(P, to_gray(sr), length, text_input)
    att = F.l1_loss(map_gt, map_pred)
    rec = weight_cross_entropy(pred, text_gt, table)
    return mse + att * 10 + rec * 0.0005, mse, att, rec, pred, map_pred, conv


# ---------------------------------------------------------------------------------------------------------------------
# text-gestalt: the stroke-focus loss (text-gestalt/loss/stroke_focus_loss.py:20-118) and its stroke-level recognizer
# (text-gestalt/loss/transformer_english_decomposition.py:8,336-398).  Same network; 10 stroke classes, the embedding /
# generator registered as *_with_upperword, a correct_list from the training branch, loss = mse + stroke_lambda * L1.
# Pinned by tests/test_text_focus.py against fixture sfl_step.npz (tools/make_golden_sfl.py).
# ---------------------------------------------------------------------------------------------------------------------
STROKES = "0123456789"                                      # transformer_english_decomposition.py:8


def stroke_schema():
    """the STT schema with 10 classes and the two renamed modules, in the reference's registration order"""
    out = OrderedDict()
    for k, v in schema(n_class=len(STROKES)).items():
        k = k.replace("embedding_word.", "embedding_word_with_upperword.").replace(
            "generator_word.", "generator_word_with_upperword.")
        out[k] = v
    return out


def make_stroke_params():
    return OrderedDict((k, v.clone()) for k, v in stroke_schema().items())


def label_stroke_encoder(labels, dic):
    """stroke_focus_loss.py:49-80: characters without a decomposition are skipped, '0' closes the word"""
    a2n = {c: i for i, c in enumerate(STROKES)}
    seqs = ["".join(dic[c] for c in s if c in dic) + "0" for s in labels]
    length = torch.tensor([len(s) for s in seqs], dtype=torch.long)
    text_input = torch.zeros(len(seqs), int(length.max()), dtype=torch.long)
    for i, s in enumerate(seqs):
        for j in range(len(s) - 1):
            text_input[i, j + 1] = a2n[s[j]]
    text_gt = torch.tensor([a2n[c] for s in seqs for c in s], dtype=torch.long)
    return length, text_input, text_gt


def stroke_recognizer(P, image, text_length, text_input):
    """transformer_english_decomposition.py:361-396 (test=False branch): 4-channel input -> luma, then the shared
    network; correct_list[i] = greedy classes of positions 0 .. L-2 reproduce the teacher-forcing input 1 .. L-1"""
    if image.shape[1] == 4:
        image = to_gray(image)
    Q = OrderedDict((k.replace("_with_upperword", ""), v) for k, v in P.items())
    pred, amap, conv = recognizer(Q, image, text_length, text_input)
    correct, start = [], 0
    for i, n in enumerate(text_length):
        n = int(n)
        correct.append(bool((pred[start:start + n].max(1)[1][:-1] == text_input[i][1:n]).all()))
        start += n
    return pred, amap, correct


def stroke_focus_loss(P, sr, hr, labels, dic, stroke_lambda=50.0):
    """stroke_focus_loss.py:83-112 (text_focus on; correct_flag is False there, so every sample takes part)"""
    mse = F.mse_loss(sr, hr)
    length, text_input, _ = label_stroke_encoder(labels, dic)
    with torch.no_grad():
        _, map_gt, correct_hr = stroke_recognizer(P, to_gray(hr), length, text_input)
    pred, map_pred, correct_sr = stroke_recognizer(P, to_gray(sr), length, text_input)
    att = F.l1_loss(map_gt, map_pred)
    return mse + att * stroke_lambda, mse, att, -1, pred, map_pred, correct_hr, correct_sr


# ---------------------------------------------------------------------------------------------------------------------
# the whole optimisation step with the text-focus / stroke-focus criterion (interfaces/super_resolution.py:69-84 with
# image_crit = TextFocusLoss, base.py:143-150): SR network of oracle/sr_oracle.py, frozen recognizer from here.  Used by
# tests/test_text_focus.py (full-size parity) and bench.py's cpu_baseline leg of the `tfl` / `sfl` configurations.
# ---------------------------------------------------------------------------------------------------------------------
def filter_labels(labels):
    """text_focus_loss.py:85: str_filt(label, 'lower') + '-' (labels of the synthetic batches are already [0-9a-z]*)"""
    return ["".join(c for c in s.lower() if c in ALPHABET[1:]) + "-" for s in labels]


def train_step_focus(P_sr, opt, P_rec, lr_img, hr_img, labels, kind="tfl", table=None, dic=None, stroke_lambda=50.0,
                     arch="tbsrn", dropout_p=0.0):
    from . import sr_oracle as O
    for p in opt.params:
        p.grad = None
    sr = O.sr_forward(P_sr, arch, lr_img, True, dropout_p=dropout_p)
    if kind == "tfl":
        loss, mse, att, rec = text_focus_loss(P_rec, sr, hr_img, filter_labels(labels), table)[:4]
    else:
        loss, mse, att, rec = stroke_focus_loss(P_rec, sr, hr_img, labels, dic, stroke_lambda)[:4]
    (loss * 100).backward()
    grads = [p.grad for p in opt.params if p.grad is not None]
    gnorm = O.clip_grad_norm(grads, 0.25)
    opt.step()
    return {"loss": float(loss.detach()), "mse": float(mse.detach()), "att": float(att.detach()),
            "rec": float(rec.detach()) if torch.is_tensor(rec) else rec, "grad_norm": float(gnorm), "sr": sr.detach()}

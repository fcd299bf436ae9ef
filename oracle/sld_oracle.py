"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Functional fp32 PyTorch-CPU restatement of the stroke-level-decomposition transformer recognizer training step
(BASELINE configs[4]).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  Every function is a pure function of a flat {state_dict key: tensor} dict `P` in the reference's schema
(tests/golden/sld_schema.json) and cites the reference file:line it follows (paths under
/root/reference/stroke-level-decomposition/).

Parity pin: tests/test_sld.py checks this restatement against fixture F7 (tools/make_golden_sld.py: the imported
reference Transformer under train.py's step, name-keyed weights): ragged predictions, cross-entropy, attention map,
encoder features, gradients, BatchNorm running statistics and a two-step Adadelta trajectory.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

ALPHABET_STROKE = "<12345$"            # util.py:14 ('<' start, '$' end, 7 classes)
LAYERS = (3, 4, 6, 3)                  # model/transformer.py:322 ResNet(num_in=3, block=BasicBlock, layers=[3,4,6,3])


# ----------------------------------------------------------------------------------------
# schema (model/transformer.py:77-125, 163-181, 184-201, 241-266, 269-296, 301-318)
# ----------------------------------------------------------------------------------------
def _conv(d, p, cout, cin, k=3):
    d[p + "weight"] = torch.zeros(cout, cin, k, k)
    d[p + "bias"] = torch.zeros(cout)


def _bn(d, p, c):
    d[p + "weight"], d[p + "bias"] = torch.ones(c), torch.zeros(c)
    d[p + "running_mean"], d[p + "running_var"] = torch.zeros(c), torch.ones(c)
    d[p + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _lin(d, p, nout, nin):
    d[p + "weight"], d[p + "bias"] = torch.zeros(nout, nin), torch.zeros(nout)


def _layer(d, p, inplanes, planes, blocks):
    for i in range(blocks):
        q = "%s%d." % (p, i)
        _conv(d, q + "conv1.", planes, inplanes if i == 0 else planes)
        _bn(d, q + "bn1.", planes)
        _conv(d, q + "conv2.", planes, planes)
        _bn(d, q + "bn2.", planes)
        if i == 0 and inplanes != planes:
            _conv(d, q + "downsample.0.", planes, inplanes)
            _bn(d, q + "downsample.1.", planes)


def _mha(d, p, dm, h):
    for i in range(4):
        _lin(d, p + "linears.%d." % i, dm, dm)
    _lin(d, p + "compress_attention_linear.", 1, h)


def schema(n_class=7, max_len=7000):
    d = OrderedDict()
    d["embedding_word.lut.weight"] = torch.zeros(n_class, 512)
    d["pe.pe"] = positional_table(512, max_len)[None]
    e = "encoder."
    _conv(d, e + "conv1.", 64, 3)
    _bn(d, e + "bn1.", 64)
    _conv(d, e + "conv2.", 128, 64)
    _bn(d, e + "bn2.", 128)
    plan = ((1, 128, 256, 256), (2, 256, 256, 256), (3, 256, 512, 512), (4, 512, 512, 1024))
    for (li, cin, planes, cout), nb in zip(plan, LAYERS):
        _layer(d, e + "layer%d." % li, cin, planes, nb)
        tail = "layer%d_conv." % li if li < 4 else "layer4_conv2."
        tbn = "layer%d_bn." % li if li < 4 else "layer4_conv2_bn."
        _conv(d, e + tail, cout, planes)
        _bn(d, e + tbn, cout)
    _mha(d, "decoder.mask_multihead.", 1024, 4)
    d["decoder.mul_layernorm1.a"], d["decoder.mul_layernorm1.b"] = torch.ones(1024), torch.zeros(1024)
    _mha(d, "decoder.multihead.", 1024, 4)
    d["decoder.mul_layernorm2.a"], d["decoder.mul_layernorm2.b"] = torch.ones(1024), torch.zeros(1024)
    _lin(d, "decoder.pff.w_1.", 2048, 1024)
    _lin(d, "decoder.pff.w_2.", 1024, 2048)
    d["decoder.mul_layernorm3.a"], d["decoder.mul_layernorm3.b"] = torch.ones(1024), torch.zeros(1024)
    _lin(d, "generator_word.proj.", n_class, 1024)
    return d


def is_buffer(k):
    return k.endswith(("running_mean", "running_var", "num_batches_tracked")) or k == "pe.pe"


def make_params(requires_grad=True):
    P = OrderedDict()
    for k, v in schema().items():
        t = v.clone()
        if requires_grad and not is_buffer(k):
            t.requires_grad_(True)
        P[k] = t
    return P


def positional_table(d_model, max_len):
    """model/transformer.py:168-176"""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


# ----------------------------------------------------------------------------------------
# label codec (util.py:90-116)
# ----------------------------------------------------------------------------------------
def converter_stroke(labels, table):
    """labels: characters; table: {char: stroke digits}.  -> (length [B], text_input [B, Lmax], text_gt [sum L])"""
    seqs = [table[s[0]] + "$" for s in labels]
    a2n = {c: i for i, c in enumerate(ALPHABET_STROKE)}
    length = torch.tensor([len(s) for s in seqs], dtype=torch.long)
    text_input = torch.zeros(len(seqs), int(length.max()), dtype=torch.long)
    for i, s in enumerate(seqs):
        for j in range(len(s) - 1):
            text_input[i, j + 1] = a2n[s[j]]
    text_gt = torch.tensor([a2n[c] for s in seqs for c in s], dtype=torch.long)
    return length, text_input, text_gt


# ----------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------
def _bnf(P, p, x, training):
    if training:
        with torch.no_grad():
            P[p + "num_batches_tracked"] += 1
    return F.batch_norm(x, P[p + "running_mean"], P[p + "running_var"], P[p + "weight"], P[p + "bias"], training,
                        0.1, 1e-5)


def _cv(P, p, x):
    return F.conv2d(x, P[p + "weight"], P[p + "bias"], padding=1)


def basic_block(P, p, x, training):
    """model/transformer.py:47-75"""
    out = F.relu(_bnf(P, p + "bn1.", _cv(P, p + "conv1.", x), training))
    out = _bnf(P, p + "bn2.", _cv(P, p + "conv2.", out), training)
    res = x
    if (p + "downsample.0.weight") in P:
        res = _bnf(P, p + "downsample.1.", _cv(P, p + "downsample.0.", x), training)
    return F.relu(out + res)


def encoder(P, x, training):
    """model/transformer.py:130-162: ONE max-pool (after conv1), then 16 x 16 maps throughout"""
    e = "encoder."
    x = F.max_pool2d(F.relu(_bnf(P, e + "bn1.", _cv(P, e + "conv1.", x), training)), 2, 2)
    x = F.relu(_bnf(P, e + "bn2.", _cv(P, e + "conv2.", x), training))
    for li, nb in zip((1, 2, 3, 4), LAYERS):
        for i in range(nb):
            x = basic_block(P, "%slayer%d.%d." % (e, li, i), x, training)
        tail = "layer%d_conv." % li if li < 4 else "layer4_conv2."
        tbn = "layer%d_bn." % li if li < 4 else "layer4_conv2_bn."
        x = F.relu(_bnf(P, e + tbn, _cv(P, e + tail, x), training))
    return x


def layernorm(x, a, b, eps=1e-6):
    """model/transformer.py:241-251: unbiased std, eps added to std"""
    return a * (x - x.mean(-1, keepdim=True)) / (x.std(-1, keepdim=True) + eps) + b


def mha(P, p, query, key, value, mask=None, dropout_p=0.0, h=4):
    """model/transformer.py:184-238 (compress_attention_linear is constructed but never used)"""
    nb, dm = query.size(0), query.size(-1)
    dk = dm // h
    q, k, v = [F.linear(t, P[p + "linears.%d.weight" % i], P[p + "linears.%d.bias" % i]).view(nb, -1, h, dk)
               .transpose(1, 2) for i, t in enumerate((query, key, value))]
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, float("-inf"))
    p_attn = F.softmax(scores, dim=-1)
    if dropout_p > 0:
        p_attn = F.dropout(p_attn, dropout_p, True)
    x = torch.matmul(p_attn, v).transpose(1, 2).contiguous().view(nb, -1, dm)
    return F.linear(x, P[p + "linears.3.weight"], P[p + "linears.3.bias"]), p_attn


def decoder(P, text, conv_feature, dropout_p=0.0):
    """model/transformer.py:285-299"""
    L = text.shape[1]
    mask = torch.tril(torch.ones(1, L, L, dtype=torch.bool))                        # subsequent_mask, :203-207
    r = layernorm(text + mha(P, "decoder.mask_multihead.", text, text, text, mask, dropout_p)[0],
                  P["decoder.mul_layernorm1.a"], P["decoder.mul_layernorm1.b"])
    b, c, hh, ww = conv_feature.shape
    mem = conv_feature.view(b, c, hh * ww).permute(0, 2, 1).contiguous()
    align, amap = mha(P, "decoder.multihead.", r, mem, mem, None, dropout_p)
    r = layernorm(r + align, P["decoder.mul_layernorm2.a"], P["decoder.mul_layernorm2.b"])
    ff = F.relu(F.linear(r, P["decoder.pff.w_1.weight"], P["decoder.pff.w_1.bias"]))
    if dropout_p > 0:
        ff = F.dropout(ff, dropout_p, True)
    ff = F.linear(ff, P["decoder.pff.w_2.weight"], P["decoder.pff.w_2.bias"])
    return layernorm(r + ff, P["decoder.mul_layernorm3.a"], P["decoder.mul_layernorm3.b"]), amap


def forward(P, image, text_length, text_input, training=True, dropout_p=0.0):
    """model/transformer.py:331-377 (train branch: ragged gather of the first `length` positions of every sample)"""
    conv = encoder(P, image, training)
    emb = F.embedding(text_input, P["embedding_word.lut.weight"]) * math.sqrt(512)      # :269-277
    pos = P["pe.pe"][:, :emb.shape[1]].expand(emb.shape[0], -1, -1)                     # pe(zeros): :178-181, :343
    if dropout_p > 0:
        pos = F.dropout(pos, dropout_p, True)
    x = torch.cat([emb, pos], 2)
    x, amap = decoder(P, x, conv, dropout_p)
    logits = F.linear(x, P["generator_word.proj.weight"], P["generator_word.proj.bias"])
    pred = torch.cat([logits[i, :int(n)] for i, n in enumerate(text_length)], 0)
    return {"pred": pred, "map": amap, "conv": conv, "logits": logits}


class AdadeltaState:
    """torch.optim.Adadelta(lr, rho, eps=1e-6) (train.py:36-38), written out"""

    def __init__(self, params, lr=1.0, rho=0.9, eps=1e-6):
        self.params, self.lr, self.rho, self.eps = list(params), lr, rho, eps
        self.sq = [torch.zeros_like(p) for p in self.params]
        self.acc = [torch.zeros_like(p) for p in self.params]

    @torch.no_grad()
    def step(self):
        for p, sq, acc in zip(self.params, self.sq, self.acc):
            if p.grad is None:
                continue
            g = p.grad
            sq.mul_(self.rho).addcmul_(g, g, value=1 - self.rho)
            delta = (acc + self.eps).sqrt() / (sq + self.eps).sqrt() * g
            acc.mul_(self.rho).addcmul_(delta, delta, value=1 - self.rho)
            p.add_(delta, alpha=-self.lr)


def train_step(P, opt, image, text_length, text_input, text_gt, dropout_p=0.0):
    """train.py:63-77"""
    for p in opt.params:
        p.grad = None
    out = forward(P, image, text_length, text_input, True, dropout_p)
    loss = F.cross_entropy(out["pred"], text_gt)
    loss.backward()
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in opt.params if p.grad is not None)))
    opt.step()
    return {"loss": float(loss), "grad_norm": gn, **{k: v.detach() for k, v in out.items()}}

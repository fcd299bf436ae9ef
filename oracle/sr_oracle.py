"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A functional fp32 PyTorch-CPU restatement of the FudanOCR scene-text-telescope /
text-gestalt hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``fudanocr_amd``) never does and fails loudly when its HIP library is missing.

Parity pin: this restatement is checked against golden vectors produced by importing the
reference's own modules in the authoring container (``tools/make_golden.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).  The CTC term is NOT in the
reference: it is ``torch.nn.functional.ctc_loss`` (third-party: PyTorch
2.10 CPU kernel) -- "parity unpinned" inside the reference, pinned by PyTorch's kernel
and additionally by the independent numpy alpha-recursion in ``oracle/ctc_numpy.py``.

Form: everything is a pure function of a flat ``{state_dict key: tensor}`` dict ``P`` whose
keys/shapes are the reference's ``state_dict`` schema, so no module
classes are shared with (or copied from) the reference.

Each function cites the reference file:line (under /root/reference/scene-text-telescope/)
that it follows.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# schema: keys, shapes, default initial values
# ----------------------------------------------------------------------------------------


def _bn(d, prefix, c):
    d[prefix + "weight"] = torch.ones(c)
    d[prefix + "bias"] = torch.zeros(c)
    d[prefix + "running_mean"] = torch.zeros(c)
    d[prefix + "running_var"] = torch.ones(c)
    d[prefix + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)


def _conv(d, prefix, cout, cin, k):
    d[prefix + "weight"] = torch.zeros(cout, cin, k, k)
    d[prefix + "bias"] = torch.zeros(cout)


def _lin(d, prefix, nout, nin):
    d[prefix + "weight"] = torch.zeros(nout, nin)
    d[prefix + "bias"] = torch.zeros(nout)


def _gru_block(d, prefix, c):
    # GruBlock: 1x1 conv + bidirectional GRU(c -> c/2)   (model/tsrn.py:128-145)
    _conv(d, prefix + "conv1.", c, c, 1)
    h = c // 2
    for suf in ("", "_reverse"):
        d[prefix + "gru.weight_ih_l0" + suf] = torch.zeros(3 * h, c)
        d[prefix + "gru.weight_hh_l0" + suf] = torch.zeros(3 * h, h)
        d[prefix + "gru.bias_ih_l0" + suf] = torch.zeros(3 * h)
        d[prefix + "gru.bias_hh_l0" + suf] = torch.zeros(3 * h)


def tps_constants(height=16, width=64, n_ctrl=20, margin=0.05):
    """Constant TPS matrices (model/tps_spatial_transformer.py:22-34,38-50,56-95)."""
    per_side = n_ctrl // 2
    xs = torch.linspace(margin, 1.0 - margin, per_side, dtype=torch.float64)
    top = torch.stack([xs, torch.full_like(xs, margin)], 1)
    bot = torch.stack([xs, torch.full_like(xs, 1.0 - margin)], 1)
    ctrl = torch.cat([top, bot], 0).float()                       # [n,2] (x,y)

    def radial(a, b):                                              # 0.5 r^2 log r^2, 0 at r=0
        diff = a[:, None, :] - b[None, :, :]
        d2 = diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]
        u = 0.5 * d2 * torch.log(d2)
        return torch.where(torch.isnan(u), torch.zeros_like(u), u)

    n = n_ctrl
    fk = torch.zeros(n + 3, n + 3)
    fk[:n, :n] = radial(ctrl, ctrl)
    fk[:n, n] = 1
    fk[n, :n] = 1
    fk[:n, n + 1:] = ctrl
    fk[n + 1:, :n] = ctrl.t()
    inv = torch.inverse(fk)
    ys, xg = torch.meshgrid(torch.arange(height, dtype=torch.float32),
                            torch.arange(width, dtype=torch.float32), indexing="ij")
    coord = torch.stack([(xg / (width - 1)).reshape(-1), (ys / (height - 1)).reshape(-1)], 1)
    rep = torch.cat([radial(coord, ctrl), torch.ones(height * width, 1), coord], 1)
    return inv, rep, ctrl


def stn_fc2_bias(n_ctrl=20, margin=0.01):
    """Initial control-point frame (model/stn_head.py:69-86)."""
    import numpy as np
    per_side = n_ctrl // 2
    xs = np.linspace(margin, 1.0 - margin, per_side)
    top = np.stack([xs, np.ones(per_side) * margin], 1)
    bot = np.stack([xs, np.ones(per_side) * (1 - margin)], 1)
    return torch.tensor(np.concatenate([top, bot], 0).astype(np.float32)).reshape(-1)


def _stn_schema(d, in_planes):
    d["tps.inverse_kernel"], d["tps.target_coordinate_repr"], ctrl = tps_constants()
    # registration order in the reference: inverse_kernel, padding_matrix, repr, ctrl points
    rep = d.pop("tps.target_coordinate_repr")
    d["tps.padding_matrix"] = torch.zeros(3, 2)
    d["tps.target_coordinate_repr"] = rep
    d["tps.target_control_points"] = ctrl
    chans = [in_planes, 32, 64, 128, 256, 256, 256]
    for i in range(6):
        p = "stn_head.stn_convnet.%d." % (2 * i)
        _conv(d, p + "0.", chans[i + 1], chans[i], 3)
        _bn(d, p + "1.", chans[i + 1])
    _lin(d, "stn_head.stn_fc1.0.", 512, 512)
    _bn(d, "stn_head.stn_fc1.1.", 512)
    _lin(d, "stn_head.stn_fc2.", 40, 512)
    d["stn_head.stn_fc2.bias"] = stn_fc2_bias()


def schema_sr(arch, in_planes=3, srb=5, stn=True, hidden=32):
    """Ordered {key: default tensor} for TSRN / TBSRN (tsrn.py:18-53, tbsrn.py:166-212)."""
    c = 2 * hidden
    d = OrderedDict()
    if arch == "tbsrn":
        _conv(d, "conv.", 3, 3, 3)        # dead layers, tbsrn.py:170-172 (input_channel=3)
        _bn(d, "bn.", 3)
    _conv(d, "block1.0.", c, in_planes, 9)
    d["block1.1.weight"] = torch.full((1,), 0.25)
    for i in range(srb):
        p = "block%d." % (i + 2)
        _conv(d, p + "conv1.", c, c, 3)
        _bn(d, p + "bn1.", c)
        _gru_block(d, p + "gru1.", c)
        _conv(d, p + "conv2.", c, c, 3)
        _bn(d, p + "bn2.", c)
        _gru_block(d, p + "gru2.", c)
        if arch == "tbsrn":
            f = p + "feature_enhancer."
            for j in range(4):
                _lin(d, f + "multihead.linears.%d." % j, 128, 128)
            _lin(d, f + "multihead.compress_attention_linear.", 1, 4)
            d[f + "mul_layernorm1.a_2"] = torch.ones(128)
            d[f + "mul_layernorm1.b_2"] = torch.zeros(128)
            _lin(d, f + "pff.w_1.", 128, 128)
            _lin(d, f + "pff.w_2.", 128, 128)
            d[f + "mul_layernorm3.a_2"] = torch.ones(128)
            d[f + "mul_layernorm3.b_2"] = torch.zeros(128)
            _lin(d, f + "linear.", 64, 128)
    p = "block%d." % (srb + 2)
    _conv(d, p + "0.", c, c, 3)
    _bn(d, p + "1.", c)
    p = "block%d." % (srb + 3)
    _conv(d, p + "0.conv.", 4 * c, c, 3)
    _conv(d, p + "1.", in_planes, c, 9)
    if stn:
        _stn_schema(d, in_planes)
    return d


def schema_crnn(nc=1, nclass=37, nh=256):
    """Ordered {key: default tensor} for CRNN(32, nc, nclass, nh) (model/crnn/crnn.py:25-68)."""
    d = OrderedDict()
    nm = [64, 128, 256, 256, 512, 512, 512]
    ks = [3, 3, 3, 3, 3, 3, 2]
    for i in range(7):
        _conv(d, "cnn.conv%d." % i, nm[i], nc if i == 0 else nm[i - 1], ks[i])
        if i in (2, 4, 6):
            _bn(d, "cnn.batchnorm%d." % i, nm[i])
    for li, (nin, nout) in enumerate(((512, nh), (nh, nclass))):
        p = "rnn.%d." % li
        for suf in ("", "_reverse"):
            d[p + "rnn.weight_ih_l0" + suf] = torch.zeros(4 * nh, nin)
            d[p + "rnn.weight_hh_l0" + suf] = torch.zeros(4 * nh, nh)
            d[p + "rnn.bias_ih_l0" + suf] = torch.zeros(4 * nh)
            d[p + "rnn.bias_hh_l0" + suf] = torch.zeros(4 * nh)
        _lin(d, p + "embedding.", nout, 2 * nh)
    return d


_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


def is_buffer(key):
    return key.rsplit(".", 1)[-1] in _BUFFER_LEAVES or key.startswith("tps.")


def dead_keys_tbsrn(key):
    """Parameters of TBSRN that never receive a gradient."""
    return (key.startswith("conv.") or key.startswith("bn.") or ".gru1." in key
            or ".gru2." in key or "compress_attention_linear" in key)


def make_params(schema, requires_grad=True):
    """Clone a schema into leaf tensors; float non-buffers get requires_grad."""
    P = OrderedDict()
    for k, v in schema.items():
        t = v.clone()
        if requires_grad and t.is_floating_point() and not is_buffer(k):
            t.requires_grad_(True)
        P[k] = t
    return P


# ----------------------------------------------------------------------------------------
# elementary pieces
# ----------------------------------------------------------------------------------------


def mish(x):
    """x * tanh(softplus(x)), softplus threshold 20 (tsrn.py:117-125)."""
    return x * torch.tanh(F.softplus(x))


def batchnorm(P, prefix, x, training, update_stats=True):
    """BatchNorm{1,2}d, torch semantics: biased var for the
    normalisation, unbiased var in the running update, momentum 0.1, eps 1e-5."""
    w, b = P[prefix + "weight"], P[prefix + "bias"]
    rm, rv = P[prefix + "running_mean"], P[prefix + "running_var"]
    dims = [0] + list(range(2, x.dim()))
    shape = [1, -1] + [1] * (x.dim() - 2)
    if training:
        mean = x.mean(dims)
        var = ((x - mean.view(shape)) ** 2).mean(dims)
        if update_stats:
            n = x.numel() // x.shape[1]
            with torch.no_grad():
                rm.mul_(0.9).add_(0.1 * mean)
                rv.mul_(0.9).add_(0.1 * var * (n / max(n - 1, 1)))
                P[prefix + "num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    xhat = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + 1e-5)
    return xhat * w.view(shape) + b.view(shape)


def conv(P, prefix, x, pad):
    return F.conv2d(x, P[prefix + "weight"], P[prefix + "bias"], padding=pad)


def linear(P, prefix, x):
    return x @ P[prefix + "weight"].t() + P[prefix + "bias"]


def layernorm_std(x, a, b, eps=1e-6):
    """The reference's own LayerNorm: unbiased std, eps added to the std (tbsrn.py:33-36)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).sum(-1, keepdim=True) / (x.shape[-1] - 1)
    return a * (x - mean) / (torch.sqrt(var) + eps) + b


def positional_encoding_2d(d_model=64, height=16, width=64):
    """Fixed 2-D sinusoid PE (tbsrn.py:39-61): first half of the channels encodes the
    column, second half the row; even channels sin, odd channels cos."""
    half = d_model // 2
    freq = torch.exp(torch.arange(0.0, half, 2) * -(math.log(10000.0) / half))   # [half/2]
    pw = torch.arange(0.0, width)[:, None] * freq[None, :]                       # [W, half/2]
    ph = torch.arange(0.0, height)[:, None] * freq[None, :]                      # [H, half/2]
    pe = torch.zeros(d_model, height, width)
    pe[0:half:2] = torch.sin(pw).t()[:, None, :].expand(-1, height, -1)
    pe[1:half:2] = torch.cos(pw).t()[:, None, :].expand(-1, height, -1)
    pe[half::2] = torch.sin(ph).t()[:, :, None].expand(-1, -1, width)
    pe[half + 1::2] = torch.cos(ph).t()[:, :, None].expand(-1, -1, width)
    return pe


def gru_bidir(P, prefix, x):
    """Bidirectional GRU, batch_first, zero initial state, gate order (r,z,n)
    (call site tsrn.py:133,141).  x: [N, T, In] -> [N, T, 2*Hid]."""
    outs = []
    for suf, rev in (("", False), ("_reverse", True)):
        wih, whh = P[prefix + "weight_ih_l0" + suf], P[prefix + "weight_hh_l0" + suf]
        bih, bhh = P[prefix + "bias_ih_l0" + suf], P[prefix + "bias_hh_l0" + suf]
        hid = whh.shape[1]
        gi_all = x @ wih.t() + bih                          # [N,T,3H]
        h = x.new_zeros(x.shape[0], hid)
        seq = [None] * x.shape[1]
        order = range(x.shape[1] - 1, -1, -1) if rev else range(x.shape[1])
        for t in order:
            gi = gi_all[:, t]
            gh = h @ whh.t() + bhh
            r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
            z = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
            n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
            h = (1 - z) * n + z * h
            seq[t] = h
        outs.append(torch.stack(seq, 1))
    return torch.cat(outs, 2)


def lstm_bidir(P, prefix, x):
    """Bidirectional LSTM, sequence-first, zero initial state, gate order (i,f,g,o)
    (call site crnn.py:11,15).  x: [T, N, In] -> [T, N, 2*Hid]."""
    outs = []
    for suf, rev in (("", False), ("_reverse", True)):
        wih, whh = P[prefix + "weight_ih_l0" + suf], P[prefix + "weight_hh_l0" + suf]
        bias = P[prefix + "bias_ih_l0" + suf] + P[prefix + "bias_hh_l0" + suf]
        hid = whh.shape[1]
        gx = x @ wih.t() + bias                              # [T,N,4H]
        h = x.new_zeros(x.shape[1], hid)
        c = x.new_zeros(x.shape[1], hid)
        seq = [None] * x.shape[0]
        order = range(x.shape[0] - 1, -1, -1) if rev else range(x.shape[0])
        for t in order:
            g = gx[t] + h @ whh.t()
            i = torch.sigmoid(g[:, :hid])
            f = torch.sigmoid(g[:, hid:2 * hid])
            gg = torch.tanh(g[:, 2 * hid:3 * hid])
            o = torch.sigmoid(g[:, 3 * hid:])
            c = f * c + i * gg
            h = o * torch.tanh(c)
            seq[t] = h
        outs.append(torch.stack(seq, 0))
    return torch.cat(outs, 2)


# ----------------------------------------------------------------------------------------
# STN head + TPS  (train mode only)
# ----------------------------------------------------------------------------------------


def stn_head(P, x, training):
    """Control-point regressor (model/stn_head.py:32-49,88-99)."""
    pools = [(2, 2), (2, 2), (2, 2), (2, 2), (1, 2), None]
    h = x
    for i in range(6):
        p = "stn_head.stn_convnet.%d." % (2 * i)
        h = F.relu(batchnorm(P, p + "1.", conv(P, p + "0.", h, 1), training))
        if pools[i] is not None:
            h = F.max_pool2d(h, pools[i], pools[i])
    feat = h.reshape(h.shape[0], -1)
    feat = F.relu(batchnorm(P, "stn_head.stn_fc1.1.", linear(P, "stn_head.stn_fc1.0.", feat),
                            training))
    pts = linear(P, "stn_head.stn_fc2.", 0.1 * feat)
    return pts.view(-1, 20, 2)


def tps_warp(P, x, ctrl):
    """TPS grid + bilinear sampling (model/tps_spatial_transformer.py:97-111,10-18);
    grid_sample modern semantics: bilinear, zero padding, align_corners=False."""
    b = ctrl.shape[0]
    y = torch.cat([ctrl, P["tps.padding_matrix"].expand(b, 3, 2)], 1)
    mapping = P["tps.inverse_kernel"] @ y
    src = P["tps.target_coordinate_repr"] @ mapping
    grid = src.view(b, x.shape[2], x.shape[3], 2).clamp(0, 1) * 2.0 - 1.0
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


# ----------------------------------------------------------------------------------------
# SR networks
# ----------------------------------------------------------------------------------------


def _gru_block_fwd(P, prefix, x):
    """GruBlock.forward (tsrn.py:135-145): 1x1 conv, rows of the map are sequences."""
    x = conv(P, prefix + "conv1.", x, 0)
    b, c, hh, ww = x.shape
    seq = x.permute(0, 2, 3, 1).reshape(b * hh, ww, c)
    out = gru_bidir(P, prefix + "gru.", seq)
    return out.view(b, hh, ww, c).permute(0, 3, 1, 2)


def _srb_front(P, p, x, training):
    r = conv(P, p + "conv1.", x, 1)
    r = mish(batchnorm(P, p + "bn1.", r, training))
    r = conv(P, p + "conv2.", r, 1)
    return batchnorm(P, p + "bn2.", r, training)


def srb_tsrn(P, p, x, training):
    """tsrn.RecurrentResidualBlock.forward (tsrn.py:89-98)."""
    r = _srb_front(P, p, x, training)
    r = _gru_block_fwd(P, p + "gru1.", r.transpose(-1, -2)).transpose(-1, -2)
    return _gru_block_fwd(P, p + "gru2.", x + r)


def attention_core(q, k, v, dropout_p=0.0):
    """softmax(q k^T / sqrt(d)) v   (tbsrn.py:132-150); dropout only when dropout_p > 0."""
    s = (q @ k.transpose(-2, -1)) / math.sqrt(q.shape[-1])
    p = torch.softmax(s, -1)
    if dropout_p > 0:
        p = F.dropout(p, dropout_p, True)
    return p @ v


def feature_enhancer(P, f, feat, dropout_p=0.0):
    """FeatureEnhancer.forward (tbsrn.py:76-92).  feat: [B,64,1024] -> [B,64,1024]."""
    b = feat.shape[0]
    pe = positional_encoding_2d(64, 16, 64).view(1, 64, 1024).expand(b, -1, -1)
    tok = torch.cat([feat, pe], 1).permute(0, 2, 1)                      # [B,1024,128]
    m = f + "multihead.linears."
    heads = lambda t: t.view(b, -1, 4, 32).transpose(1, 2)
    q, k, v = (heads(linear(P, m + "%d." % j, tok)) for j in range(3))
    att = attention_core(q, k, v, dropout_p).transpose(1, 2).reshape(b, -1, 128)
    att = linear(P, m + "3.", att)
    r = layernorm_std(tok + att, P[f + "mul_layernorm1.a_2"], P[f + "mul_layernorm1.b_2"])
    ff = F.relu(linear(P, f + "pff.w_1.", r))
    if dropout_p > 0:
        ff = F.dropout(ff, dropout_p, True)
    ff = linear(P, f + "pff.w_2.", ff)
    r = layernorm_std(r + ff, P[f + "mul_layernorm3.a_2"], P[f + "mul_layernorm3.b_2"])
    return linear(P, f + "linear.", r).permute(0, 2, 1)


def srb_tbsrn(P, p, x, training, dropout_p=0.0):
    """tbsrn.RecurrentResidualBlock.forward (tbsrn.py:246-257)."""
    r = _srb_front(P, p, x, training)
    s = r.shape
    r = feature_enhancer(P, p + "feature_enhancer.", r.reshape(s[0], s[1], -1), dropout_p)
    return x + r.reshape(s)


def sr_forward(P, arch, x, training, srb=5, stn=True, dropout_p=0.0):
    """TSRN.forward / TBSRN.forward (tsrn.py:60-74, tbsrn.py:214-226)."""
    if stn and training:
        x = tps_warp(P, x, stn_head(P, x, training))
    b1 = conv(P, "block1.0.", x, 4)
    b1 = torch.where(b1 >= 0, b1, P["block1.1.weight"] * b1)           # PReLU, single slope
    h = b1
    for i in range(srb):
        p = "block%d." % (i + 2)
        h = srb_tsrn(P, p, h, training) if arch == "tsrn" else srb_tbsrn(P, p, h, training,
                                                                        dropout_p)
    p = "block%d." % (srb + 2)
    h = batchnorm(P, p + "1.", conv(P, p + "0.", h, 1), training)
    p = "block%d." % (srb + 3)
    u = mish(F.pixel_shuffle(conv(P, p + "0.conv.", b1 + h, 1), 2))
    return torch.tanh(conv(P, p + "1.", u, 4))


# ----------------------------------------------------------------------------------------
# recognizer leg
# ----------------------------------------------------------------------------------------


def parse_crnn_data(img):
    """TextBase.parse_crnn_data (interfaces/base.py:319-325): bicubic to 32x100 + luma."""
    g = F.interpolate(img, (32, 100), mode="bicubic", align_corners=False)
    return 0.299 * g[:, 0:1] + 0.587 * g[:, 1:2] + 0.114 * g[:, 2:3]


def crnn_forward(P, x, training=False):
    """CRNN.forward (model/crnn/crnn.py:31-80).  x [B,1,32,100] -> logits [26,B,37]."""
    pads = [1, 1, 1, 1, 1, 1, 0]
    h = x
    for i in range(7):
        h = conv(P, "cnn.conv%d." % i, h, pads[i])
        if i in (2, 4, 6):
            h = batchnorm(P, "cnn.batchnorm%d." % i, h, training)
        h = F.relu(h)
        if i in (0, 1):
            h = F.max_pool2d(h, 2, 2)
        elif i in (3, 5):
            h = F.max_pool2d(h, (2, 2), (2, 1), (0, 1))
    assert h.shape[2] == 1
    seq = h.squeeze(2).permute(2, 0, 1)                                   # [W,B,C]
    for li in range(2):
        p = "rnn.%d." % li
        rec = lstm_bidir(P, p + "rnn.", seq)
        t, b, hh = rec.shape
        seq = linear(P, p + "embedding.", rec.reshape(t * b, hh)).view(t, b, -1)
    return seq


ALPHABET = "0123456789abcdefghijklmnopqrstuvwxyz"


def encode_labels(strs):
    """utils/utils_crnn.py:32-52 : blank = 0, '0-9a-z' -> 1..36, case-insensitive."""
    flat, lens = [], []
    for s in strs:
        lens.append(len(s))
        flat.extend(ALPHABET.index(ch.lower()) + 1 for ch in s)
    return torch.tensor(flat, dtype=torch.int32), torch.tensor(lens, dtype=torch.int32)


def greedy_decode(logits_tbc):
    """get_crnn_pred (interfaces/super_resolution.py:143-158): argmax, collapse, drop blank."""
    alpha = "-" + ALPHABET
    idx = logits_tbc.permute(1, 0, 2).argmax(2)
    out = []
    for row in idx.tolist():
        s, last = "", 0
        for i in row:
            if i != 0 and i != last:
                s += alpha[i]
            last = i
        out.append(s)
    return out


def ctc_from_logits(logits_tbc, targets, target_lengths):
    """Build-defined CTC term: log_softmax + F.ctc_loss, blank 0,
    reduction 'mean' (per-sample / target length, then batch mean), zero_infinity=True."""
    t, b, _ = logits_tbc.shape
    lp = F.log_softmax(logits_tbc, 2)
    return F.ctc_loss(lp, targets.long(), torch.full((b,), t, dtype=torch.long),
                      target_lengths.long(), blank=0, reduction="mean", zero_infinity=True)


# ----------------------------------------------------------------------------------------
# the measured step
# ----------------------------------------------------------------------------------------


def step_loss(P, arch, lr_img, hr_img, crnn_P=None, targets=None, target_lengths=None,
              training=True, dropout_p=0.0, srb=5, stn=True):
    """SR forward + MSE (+ CRNN -> CTC)   (SURVEY.md section 3.3).  Returns (loss, mse, ctc, sr)."""
    sr = sr_forward(P, arch, lr_img, training, srb, stn, dropout_p)
    mse = ((sr - hr_img) ** 2).mean()                                    # text_focus_loss.py:86
    ctc = None
    loss = mse
    if crnn_P is not None:
        logits = crnn_forward(crnn_P, parse_crnn_data(sr[:, :3]), training=False)
        ctc = ctc_from_logits(logits, targets, target_lengths)
        loss = mse + ctc
    return loss, mse, ctc, sr


def clip_grad_norm(grads, max_norm=0.25):
    """torch.nn.utils.clip_grad_norm_ semantics. Returns total norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in grads:
            g.mul_(coef)
    return total


class AdamState:
    """Adam(lr 1e-4, betas (0.5, 0.999), eps 1e-8), bias-corrected (interfaces/base.py:194-198)."""

    def __init__(self, params, lr=1e-4, beta1=0.5, beta2=0.999, eps=1e-8):
        self.params = params
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        c1 = 1 - self.b1 ** self.t
        c2 = 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(c2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / c1)


def train_step(P, opt, arch, lr_img, hr_img, crnn_P=None, targets=None, target_lengths=None,
               dropout_p=0.0, srb=5, stn=True):
    """One optimisation step (interfaces/super_resolution.py:69-84): loss*100, backward,
    clip 0.25 on the params that have grads, Adam.  Returns dict of scalars."""
    for p in opt.params:
        p.grad = None
    loss, mse, ctc, sr = step_loss(P, arch, lr_img, hr_img, crnn_P, targets, target_lengths,
                                   True, dropout_p, srb, stn)
    (loss * 100).backward()
    grads = [p.grad for p in opt.params if p.grad is not None]
    gnorm = clip_grad_norm(grads, 0.25)
    opt.step()
    return {"loss": float(loss.detach()), "mse": float(mse.detach()), "ctc": None if ctc is None else float(ctc.detach()),
            "grad_norm": float(gnorm), "sr": sr.detach()}

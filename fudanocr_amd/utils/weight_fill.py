"""Name-keyed deterministic parameter fill.

Every tensor of a ``state_dict`` is filled from a generator seeded by the CRC32 of its
key, so the reference model (in the authoring container), the CPU oracle and the HIP
product model all get bit-identical weights from the key names alone -- weights never
have to travel as fixtures.

Rules (chosen so that every layer is numerically "alive"):
  * ``*.num_batches_tracked``, ``stn_head.stn_fc2.bias`` : untouched (the fc2 bias is the control-point
    frame, reference stn_head.py:69-86).
  * ``tps.*`` buffers: the TPS constants recomputed in FLOAT64 and cast once (``tps_buffers``).  The
    reference builds them with an fp32 LAPACK inverse at construction time, which differs between CPUs
    by up to 9e-4 absolute (measured: this container vs the GPU box's EPYC) -- enough to move SR pixels
    by 1e-3 on noise images.  They are state_dict entries, so overwriting them is a plain load.
  * BatchNorm (a key that has a sibling ``running_mean``): weight U[0.5,1.5],
    bias U[-0.1,0.1], running_mean U[-0.1,0.1], running_var U[0.5,1.5].
  * LayerNorm ``a_2`` U[0.5,1.5], ``b_2`` U[-0.1,0.1]  (reference tbsrn.py:23-36); the stroke-level-decomposition
    transformer names them ``a`` / ``b`` (SLD model/transformer.py:247-248): same ranges.
  * ``pe.pe`` (the SLD positional-encoding buffer, a state_dict entry): untouched.
  * PReLU slope (shape [1] weight): U[0.1,0.4].
  * dim >= 2: U[-1/sqrt(fan_in), +1/sqrt(fan_in)], fan_in = numel / shape[0];
    ``stn_head.stn_fc2.weight`` additionally scaled by 0.2 (keeps the warp mild).
  * remaining 1-D tensors (biases): U[-0.1, 0.1].
"""
import math
import zlib

import torch


def _u(shape, lo, hi, key):
    """Uniform[lo, hi) on a 2^24 grid, BIT-IDENTICAL on every host: integer draws (mt19937), then one
    correctly rounded float64 multiply and an exact cast.  (`rand()*(hi-lo)+lo` is not: a fused
    multiply-add on one CPU vs mul+add on another moves values by 1 ulp, which this network's B=4
    batch-norm chain amplifies to ~1e-3 on the SR pixels.)"""
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    k = torch.randint(0, 1 << 24, tuple(shape), generator=g, dtype=torch.int64)
    step = (hi - lo) / float(1 << 24)
    off = int(round(lo / step))                 # range start snapped to the grid (exact for symmetric ranges)
    return ((k + off).to(torch.float64) * step).to(torch.float32)


_TPS_CACHE = {}


def tps_buffers(height=16, width=64, n_ctrl=20, margin=0.05):
    """TPS constants (reference model/tps_spatial_transformer.py:56-95) in float64, cast to fp32:
    inverse of the padded kernel matrix, the per-pixel basis [U(|p - c_j|), 1, x, y], control frame."""
    key = (height, width, n_ctrl, margin)
    if key in _TPS_CACHE:
        return _TPS_CACHE[key]
    half = n_ctrl // 2
    xs = torch.linspace(margin, 1.0 - margin, half, dtype=torch.float64)
    ctrl = torch.cat([torch.stack([xs, torch.full_like(xs, margin)], 1),
                      torch.stack([xs, torch.full_like(xs, 1.0 - margin)], 1)], 0).float().double()

    def basis(a, b):
        d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        return torch.where(d2 > 0, 0.5 * d2 * torch.log(d2.clamp_min(1e-300)), torch.zeros_like(d2))

    n = n_ctrl
    sysm = torch.zeros(n + 3, n + 3, dtype=torch.float64)
    sysm[:n, :n] = basis(ctrl, ctrl)
    sysm[:n, n] = 1
    sysm[n, :n] = 1
    sysm[:n, n + 1:] = ctrl
    sysm[n + 1:, :n] = ctrl.t()
    gy, gx = torch.meshgrid(torch.arange(height, dtype=torch.float64), torch.arange(width, dtype=torch.float64),
                            indexing="ij")
    xy = torch.stack([gx.reshape(-1) / (width - 1), gy.reshape(-1) / (height - 1)], 1)
    rep = torch.cat([basis(xy, ctrl), torch.ones(height * width, 1, dtype=torch.float64), xy], 1)
    out = {"inverse_kernel": torch.linalg.inv(sysm).float().contiguous(), "padding_matrix": torch.zeros(3, 2),
           "target_coordinate_repr": rep.float().contiguous(), "target_control_points": ctrl.float()}
    _TPS_CACHE[key] = out
    return out


def fill_value(key, shape, siblings):
    """Return the fp32 CPU tensor for `key`, or None when the entry is left untouched."""
    leaf = key.rsplit(".", 1)[-1]
    prefix = key[: -len(leaf)]
    if key.startswith("tps."):
        v = tps_buffers().get(leaf)
        return v.clone() if v is not None and tuple(v.shape) == tuple(shape) else None
    if leaf == "num_batches_tracked":
        return None
    if key == "stn_head.stn_fc2.bias":
        return None
    is_bn = (prefix + "running_mean") in siblings
    if is_bn:
        if leaf == "weight":
            return _u(shape, 0.5, 1.5, key)
        if leaf == "bias":
            return _u(shape, -0.1, 0.1, key)
        if leaf == "running_mean":
            return _u(shape, -0.1, 0.1, key)
        if leaf == "running_var":
            return _u(shape, 0.5, 1.5, key)
        return None
    if leaf == "pe" and len(shape) == 3:
        return None
    if leaf == "a_2" or (leaf == "a" and len(shape) == 1):
        return _u(shape, 0.5, 1.5, key)
    if leaf == "b_2":
        return _u(shape, -0.1, 0.1, key)
    if len(shape) >= 2:
        numel = 1
        for s in shape:
            numel *= s
        fan_in = max(1, numel // shape[0])
        b = 1.0 / math.sqrt(fan_in)
        t = _u(shape, -b, b, key)
        if key == "stn_head.stn_fc2.weight":
            t = t * 0.2
        return t
    if leaf == "weight" and tuple(shape) == (1,):
        return _u(shape, 0.1, 0.4, key)
    return _u(shape, -0.1, 0.1, key)


@torch.no_grad()
def fill_module_(module):
    """Fill every entry of ``module.state_dict()`` in place; returns the module."""
    sd = module.state_dict()
    keys = set(sd.keys())
    for k in sorted(sd.keys()):
        v = fill_value(k, tuple(sd[k].shape), keys)
        if v is not None:
            sd[k].copy_(v.to(sd[k].device))
    return module


@torch.no_grad()
def fill_dict_(params):
    """Same rule for a plain ``{key: tensor}`` dict (used by the functional oracle)."""
    keys = set(params.keys())
    for k in sorted(params.keys()):
        v = fill_value(k, tuple(params[k].shape), keys)
        if v is not None:
            params[k].copy_(v)
    return params

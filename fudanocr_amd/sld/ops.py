"""autograd wrappers over the C-ABI entry points that only the stroke-level-decomposition recognizer needs
(csrc/sld_ops.hip; include/focr.h, last section).  Everything else (convolutions on the halo kernel, BatchNorm with
epilogue statistics, linears, LayerNorm, dropout, max-pool) comes from fudanocr_amd.kernels."""
import ctypes
import math

import torch

from .. import _lib
from .. import kernels as K

_p, _chk, _stream, _target = K._p, K._chk, K._stream, K._target


def _ip(t):
    return ctypes.c_void_p(t.data_ptr())


class _AddReLU(torch.autograd.Function):
    """relu(a + b) (BasicBlock tail, reference model/transformer.py:66-75)"""

    @staticmethod
    def forward(ctx, a, b):
        _chk(a, b)
        y = torch.empty_like(a)
        _lib.call("focr_add_relu_fwd", _p(a), _p(b), _p(y), a.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        g = torch.empty_like(dy)
        _lib.call("focr_relu_bwd", _p(dy), _p(y), _p(g), dy.numel(), _stream())
        return g, g


def add_relu(a, b):
    return _AddReLU.apply(a, b)


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, idx, table, scale):
        idx = idx.contiguous()
        if not (idx.is_cuda and idx.dtype == torch.int64):
            raise RuntimeError("embedding needs int64 CUDA indices")
        _chk(table)
        rows, d = idx.numel(), table.shape[1]
        y = torch.empty(tuple(idx.shape) + (d,), device=table.device, dtype=torch.float32)
        _lib.call("focr_embedding_fwd", _ip(idx), _p(table), _p(y), rows, d, float(scale), _stream())
        ctx.cfg = (rows, d, float(scale), tuple(table.shape))
        ctx.target = _target(table)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        rows, d, scale, tshape = ctx.cfg
        dy = dy.contiguous()
        dt = ctx.target if ctx.target is not None else torch.zeros(tshape, device=dy.device, dtype=torch.float32)
        _lib.call("focr_embedding_bwd", _ip(idx), _p(dy), _p(dt), rows, d, scale, _stream())
        return None, (None if ctx.target is not None else dt), None


def embedding(idx, table, scale):
    return _Embedding.apply(idx, table, scale)


class _GatherRows(torch.autograd.Function):
    """out[r] = x2d[idx[r]] with unique indices (the ragged prediction gather, transformer.py:362-370)"""

    @staticmethod
    def forward(ctx, x2d, idx):
        _chk(x2d)
        out = torch.empty((idx.numel(), x2d.shape[1]), device=x2d.device, dtype=torch.float32)
        _lib.call("focr_gather_rows", _p(x2d), _ip(idx), _p(out), idx.numel(), x2d.shape[1], 0, _stream())
        ctx.shape = tuple(x2d.shape)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.zeros(ctx.shape, device=dout.device, dtype=torch.float32)
        _lib.call("focr_gather_rows", _p(dout), _ip(idx), _p(dx), idx.numel(), ctx.shape[1], 1, _stream())
        return dx, None


def gather_rows(x2d, idx):
    return _GatherRows.apply(x2d, idx)


class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss (mean) on [rows, C <= 64] logits, int64 targets"""

    @staticmethod
    def forward(ctx, logits, target):
        logits = logits.contiguous()
        _chk(logits)
        rows, c = logits.shape
        loss = torch.empty(1, device=logits.device)
        ws = torch.empty(rows, device=logits.device)
        grad = torch.empty_like(logits)
        _lib.call("focr_cross_entropy_fwd", _p(logits), _ip(target.contiguous()), _p(loss), _p(ws), _p(grad), rows, c,
                  _stream())
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        _lib.call("focr_scale_dev", _p(grad), _p(g.contiguous().reshape(1)), _p(out), grad.numel(), _stream())
        return out, None


def cross_entropy(logits, target):
    return _CrossEntropy.apply(logits, target)


class _SmallAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(d_k)) [causal mask] [dropout] v for heads of 256 inside [B, L, H*256] row layouts; returns
    (o, attention map after dropout)"""

    @staticmethod
    def forward(ctx, q, k, v, heads, causal, p_drop, seed):
        _chk(q, k, v)
        b, lq, dm = q.shape
        lk = k.shape[1]
        dk = dm // heads
        o = torch.empty_like(q)
        p = torch.empty((b, heads, lq, lk), device=q.device, dtype=torch.float32)
        pd = torch.empty_like(p)
        scale = 1.0 / math.sqrt(dk)
        _lib.call("focr_small_attention_fwd", _p(q), _p(k), _p(v), _p(o), _p(p), _p(pd), b, heads, lq, lk, dk, dm, dm, dm,
                  scale, int(causal), float(p_drop), seed, _stream())
        ctx.cfg = (b, heads, lq, lk, dk, dm, scale, int(causal))
        ctx.save_for_backward(q, k, v, p, pd)
        return o, pd

    @staticmethod
    def backward(ctx, do, dmap):
        """dmap: gradient arriving through the returned attention map (the text-focus loss' L1 term); None otherwise"""
        q, k, v, p, pd = ctx.saved_tensors
        b, heads, lq, lk, dk, dm, scale, causal = ctx.cfg
        do = torch.zeros_like(q) if do is None else do.contiguous()
        dmap = None if dmap is None else dmap.contiguous()
        dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = torch.empty_like(p)
        _lib.call("focr_small_attention_bwd", _p(q), _p(k), _p(v), _p(do), _p(p), _p(pd), _p(dmap), _p(dq), _p(dk_),
                  _p(dv), _p(ws), b, heads, lq, lk, dk, dm, dm, dm, scale, causal, _stream())
        return dq, dk_, dv, None, None, None, None


def small_attention(q, k, v, heads=4, causal=False, p_drop=0.0):
    seed = K._new_seed() if p_drop > 0 else 0
    return _SmallAttention.apply(q, k, v, heads, causal, p_drop, seed)


def adadelta(p, g, sq, acc, lr, rho, eps, gscale=1.0):
    _lib.call("focr_adadelta", _p(p), _p(g), _p(sq), _p(acc), p.numel(), float(lr), float(rho), float(eps),
              float(gscale), _stream())


class _L1(torch.autograd.Function):
    """nn.L1Loss(mean) with the gradient flowing to the second argument only (the first is the no-grad HR map)"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        out = torch.empty(1, device=a.device)
        ws = torch.empty(256, device=a.device)
        _lib.call("focr_l1_fwd", _p(a), _p(b), _p(out), _p(ws), a.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        db = torch.empty_like(b)
        _lib.call("focr_l1_bwd", _p(a), _p(b), _p(g.contiguous().reshape(1)), _p(db), b.numel(), _stream())
        return None, db


def l1_loss(a_const, b):
    return _L1.apply(a_const, b)


class _WeightCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, table):
        logits = logits.contiguous()
        _chk(logits, table)
        rows, c = logits.shape
        loss = torch.empty(1, device=logits.device)
        ws = torch.empty(rows, device=logits.device)
        grad = torch.empty_like(logits)
        _lib.call("focr_weight_cross_entropy_fwd", _p(logits), _ip(target.contiguous()), _p(table), _p(loss), _p(ws),
                  _p(grad), rows, c, _stream())
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        _lib.call("focr_scale_dev", _p(grad), _p(g.contiguous().reshape(1)), _p(out), grad.numel(), _stream())
        return out, None, None


def weight_cross_entropy(logits, target, table):
    """loss/weight_ce_loss.py:38-45 with the [C, C] weight table as an argument"""
    return _WeightCE.apply(logits, target, table)


class _L1Masked(torch.autograd.Function):
    """l1_loss on padded maps [outer, L, inner]: positions j >= plan[0] (device int64) are outside the sum, the mean and the
    gradient (csrc/sld_ops.hip: the recordable form of the focus losses)"""

    @staticmethod
    def forward(ctx, a, b, plan):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        inner, length = a.shape[-1], a.shape[-2]
        outer = a.numel() // (inner * length)
        out = torch.empty(1, device=a.device)
        ws = torch.empty(256, device=a.device)
        _lib.call("focr_l1_masked_fwd", _p(a), _p(b), _p(out), _p(ws), outer, length, inner, _ip(plan), _stream())
        ctx.save_for_backward(a, b, plan)
        ctx.cfg = (outer, length, inner)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b, plan = ctx.saved_tensors
        outer, length, inner = ctx.cfg
        db = torch.empty_like(b)
        _lib.call("focr_l1_masked_bwd", _p(a), _p(b), _p(g.contiguous().reshape(1)), _p(db), outer, length, inner, _ip(plan),
                  _stream())
        return None, db, None


def l1_loss_masked(a_const, b, plan):
    return _L1Masked.apply(a_const, b, plan)


class _WeightCEMasked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, table, plan):
        logits = logits.contiguous()
        _chk(logits, table)
        rows, c = logits.shape
        loss = torch.empty(1, device=logits.device)
        ws = torch.empty(rows, device=logits.device)
        grad = torch.empty_like(logits)
        _lib.call("focr_weight_cross_entropy_masked_fwd", _p(logits), _ip(target), _p(table), _p(loss), _p(ws), _p(grad), rows,
                  c, _ip(plan), _stream())
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        out = torch.empty_like(grad)
        _lib.call("focr_scale_dev", _p(grad), _p(g.contiguous().reshape(1)), _p(out), grad.numel(), _stream())
        return out, None, None, None


def weight_cross_entropy_masked(logits, target_padded, table, plan):
    """weight_cross_entropy over padded rows [B * L, C]: target < 0 = padding; mean over plan[1] (device int64) rows"""
    return _WeightCEMasked.apply(logits, target_padded, table, plan)

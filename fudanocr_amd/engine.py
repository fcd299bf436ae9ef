"""Optimisation-step engine: the reference's train-step tail
(interfaces/super_resolution.py:79-84: loss*100 -> backward -> clip_grad_norm_(0.25) -> Adam)
restated for one process per GPU.

* All trainable parameters live in ONE flat fp32 buffer (parameters become strided views of it,
  keeping their state_dict shapes), their gradients in a second one.  Dead parameters
   simply keep a zero gradient: with zero first/second moments Adam leaves
  them untouched, exactly like torch skipping `grad is None`.
* Data parallel (replaces nn.DataParallel, base.py:178-179): the flat gradient buffer is
  all-reduced (RCCL via torch.distributed, SUM) in a few large buckets; the 1/world averaging is
  folded into the fused norm / clip+Adam kernels.  BatchNorm statistics stay per shard, as in
  DataParallel.  No parameter broadcast per step: replicas apply identical averaged gradients.
* clip + Adam: two HIP kernels over the flat buffers, clip coefficient computed on the device,
  no host synchronisation in the step.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import kernels as K


def _physical(p):
    """(shape, permutation) such that p.permute(perm) is contiguous (dense tensors only)."""
    order = sorted(range(p.dim()), key=lambda d: (-p.stride(d), d))
    return [p.shape[d] for d in order], order


def _qkv_adjacent_order(model):
    """model.parameters() with the q/k/v projection weights (then biases) of every self-attention module made
    consecutive, so that they can be used as one packed [3*d, d] operand straight from the flat buffers."""
    params = list(model.named_parameters())
    by_name = dict(params)
    out, done = [], set()
    for name, p in params:
        if name in done:
            continue
        if name.endswith(".linears.0.weight"):
            stem = name[:-len("0.weight")]
            group = [stem + "%d.weight" % i for i in range(3)] + [stem + "%d.bias" % i for i in range(3)]
            if all(g in by_name for g in group):
                for g in group:
                    out.append(by_name[g])
                    done.add(g)
                continue
        if name.endswith(".weight_ih_l0") and (name[:-len("weight_ih_l0")] + "weight_hh_l0_reverse") in by_name:
            # bidirectional GRU / LSTM registry (TSRN GruBlock): forward and reverse tensors of each kind side by side,
            # so [W_ih | W_ih_reverse], stacked W_hh and the biases are views of the flat buffers too
            stem = name[:-len("weight_ih_l0")]
            group = [stem + k + sfx for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
                     for sfx in ("", "_reverse")]
            if all(g in by_name for g in group):
                for g in group:
                    out.append(by_name[g])
                    done.add(g)
                continue
        out.append(p)
        done.add(name)
    return out


class FlatBuffers:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        self.offsets, total = [], 0
        for p in self.params:
            self.offsets.append(total)
            total += (p.numel() + 3) // 4 * 4              # keep every slice 16-byte aligned
        self.numel = total
        self.flat_param = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                shape, order = _physical(p)
                inv = [order.index(d) for d in range(p.dim())]
                view = self.flat_param[off:off + p.numel()].view(shape).permute(inv) if p.dim() else \
                    self.flat_param[off:off + 1].view(())
                view.copy_(p.data)
                p.data = view
                g = self.flat_grad[off:off + p.numel()].view(shape).permute(inv) if p.dim() else \
                    self.flat_grad[off:off + 1].view(())
                p.grad = g
                p._focr_grad = g          # kernels write the gradient here directly (kernels._target)

    def zero_grad(self):
        if self.flat_grad.is_cuda:
            from . import _lib
            _lib.call("focr_zero", K._p(self.flat_grad), self.flat_grad.numel(), K._stream())
        else:
            self.flat_grad.zero_()
        for p, off in zip(self.params, self.offsets):      # re-attach if something reset .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                shape, order = _physical(p)
                inv = [order.index(d) for d in range(p.dim())]
                p.grad = self.flat_grad[off:off + p.numel()].view(shape).permute(inv) if p.dim() else \
                    self.flat_grad[off:off + 1].view(())
                p._focr_grad = p.grad


class FusedClipAdam:
    """clip_grad_norm_(max_norm) + Adam on flat buffers (HIP on CUDA tensors).  `world` folds the
    data-parallel averaging into the kernels."""

    def __init__(self, flat, lr=1e-4, betas=(0.5, 0.999), eps=1e-8, max_norm=0.25, state=None):
        self.flat, self.lr, self.betas, self.eps, self.max_norm = flat, lr, betas, eps, max_norm
        self.m = torch.zeros_like(flat.flat_param)
        self.v = torch.zeros_like(flat.flat_param)
        self.sumsq = K.sumsq_workspace(flat.flat_param.device)
        self.t = 0
        self.state = state        # kernels.StepState: step count / bias corrections live on the device (the owner advances it)

    def step(self, world=1):
        self.t += 1
        g = 1.0 / world
        K.grad_sumsq(self.flat.flat_grad, self.sumsq, g)
        if self.state is not None:
            K.clip_adam_state(self.flat.flat_param, self.flat.flat_grad, self.m, self.v, self.sumsq, self.lr,
                              self.betas[0], self.betas[1], self.eps, self.state.buf, self.max_norm, g)
        else:
            K.clip_adam(self.flat.flat_param, self.flat.flat_grad, self.m, self.v, self.sumsq, self.lr,
                        self.betas[0], self.betas[1], self.eps, self.t, self.max_norm, g)

    def grad_norm(self):
        return self.sumsq[:1].sqrt()


class _Boundary(torch.autograd.Function):
    """Identity whose backward fires a callback: when the gradient flows back past this point every
    parameter gradient of the layers behind it is final, so their bucket can start its all-reduce."""

    @staticmethod
    def forward(ctx, x, cb):
        ctx.cb = cb
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.cb()
        return g, None


_HR_SIDE = os.environ.get("FOCR_HR_SIDE", "1") != "0"      # the focus losses' HR branch on the side stream, under the SR forward


class TrainStep:
    """model: SR net (TBSRN/TSRN); crit: CTCFocusLoss.  One call = one optimisation step.

    Data-parallel overlap: `boundaries` names top-level sub-modules (in forward order).  The flat
    gradient range of everything from a boundary up to `tail` (modules whose gradients arrive last,
    i.e. the STN head that sits in front of the network) is all-reduced on a side stream as soon as
    backward has passed that boundary; the remainder goes out after backward.  RCCL runs on its own
    stream, so these collectives overlap the remaining backward kernels."""

    REPLAY_WARMUP = 2      # eager steps per input signature before the step is recorded (lazy tables, workspaces)
    MAX_RECORDINGS = 6     # signatures kept (each recording owns a private memory pool of the step's temporaries): e.g. the
    #                        full batch, an epoch's short last batch, a second precision mode -- or, with the focus losses, the
    #                        label-capacity buckets of the full batch (text-focus: multiples of 4 symbols); further signatures step eagerly

    def __init__(self, model, crit, lr=1e-4, betas=(0.5, 0.999), max_norm=0.25, process_group=None,
                 wgrad_side_stream=True, n_buckets=4, dropout=True, boundaries=("block3", "block6"),
                 tail=("stn_head",), replay=None, seed=None):
        self.model, self.crit = model, crit
        self.dropout = dropout        # False: nn.Dropout slots stay in eval (parity runs)
        self.flat = FlatBuffers(_qkv_adjacent_order(model))
        self._attach_packed_qkv()
        self._attach_packed_gru()
        on_gpu = self.flat.flat_grad.is_cuda
        # per-step scalars (dropout epoch, Adam step count + bias corrections) live in device memory and are advanced by the
        # first launch of every step: the host passes the same arguments in every step, which is what lets a step be
        # RECORDED once and re-issued from C (replay.py, csrc/replay.hip)
        self.state = K.StepState(self.flat.flat_grad.device, betas[0], betas[1]) if on_gpu else None
        self.opt = FusedClipAdam(self.flat, lr, betas, 1e-8, max_norm, state=self.state)
        # replay: None -> FOCR_REPLAY (default on).  Effective on one GPU per process group of size 1 with a criterion that
        # declares REPLAY_SAFE (no host-side work per step beyond label encoding); everything else steps eagerly.
        self.replay = (os.environ.get("FOCR_REPLAY", "1") != "0") if replay is None else bool(replay)
        self._recs, self._seen, self.recorded = {}, {}, None
        self.pg = process_group
        self.wgrad_side_stream = bool(wgrad_side_stream)
        self.flips = K.FlipTable()               # one batched weight flip per step for all data-gradient GEMMs
        self.frags = K.FragTable(managed=True)   # pre-split fragment-ordered weights of the halo-kernel layers: one
                                                 # batched preparation launch per step (forward + data-gradient forms)
        self.ctx = K.StepContext()
        # dropout: fixed per-site seeds + the device-resident epoch (same keep bits whether a step is launched eagerly or
        # replayed; `seed` makes runs repeatable)
        self.ctx.seed_base = (int(seed) if seed is not None else
                              int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())) if on_gpu else None
        self.ctx.mask_prefetch = True            # attention keep bits of step k + 1 are drawn on the side stream
        #                                          this engine's deferred gradients / tables / side stream (the autograd
                                                 # nodes recorded during its forward carry it into their backward)
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.world > 1:
            # replicas must START identical (nn.DataParallel replicates device 0's module, base.py:178-179); after this
            # one broadcast they stay bit-identical without any per-step parameter traffic
            dist.broadcast(self.flat.flat_param, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                           group=process_group)
        # FOCR_COMM=native: the gradient ranges go through the library's own RCCL communicator (include/focr.h
        # focr_comm_init / focr_allreduce_async) instead of torch.distributed's; the process group is then only the host
        # channel for the 128-byte unique id and the initial parameter broadcast
        self.native_comm = False
        if self.world > 1 and os.environ.get("FOCR_COMM", "") == "native" and self.flat.flat_grad.is_cuda:
            from . import _lib
            ident = [None]
            if dist.get_rank(process_group) == 0:
                buf = ctypes.create_string_buffer(128)
                _lib.call("focr_comm_unique_id", buf)
                ident[0] = bytes(buf.raw)
            dist.broadcast_object_list(ident, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                                       group=process_group)
            _lib.call("focr_comm_init", dist.get_rank(process_group), self.world, ctypes.c_char_p(ident[0]))
            self.native_comm = True
        n = self.flat.numel
        edges = [n * i // n_buckets // 4 * 4 for i in range(n_buckets)] + [n]
        self.buckets = [(edges[i], edges[i + 1]) for i in range(n_buckets) if edges[i + 1] > edges[i]]
        self._works, self._sent = [], []
        # bench.py's N > 1 line: per step, the time the COMPUTE stream spends waiting for the exchange after the last
        # backward / weight-gradient kernel (event pair around the tail launches + waits) = the exposed communication
        self.comm_timing, self._comm_events = False, []
        self._g100 = None
        self._plan_overlap(boundaries, tail)

    def _attach_packed_qkv(self):
        """Self-attention modules project q, k, v with one packed [3*d, d] GEMM.  Their three weights (and biases)
        sit next to each other in the flat buffers (_qkv_adjacent_order), so the packed operand is a VIEW of the flat
        parameter buffer and its gradient target a view of the flat gradient: no torch.cat per forward, no slice
        gradient accumulation kernels per backward (30 tiny adds + 10 cats per step on TBSRN)."""
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        for m in self.model.modules():
            lin = getattr(m, "linears", None)
            if lin is None or not hasattr(m, "_packed_qkv") or len(lin) < 3:
                continue
            ws, bs = [lin[i].weight for i in range(3)], [lin[i].bias for i in range(3)]
            if any(id(t) not in off for t in ws + bs):
                continue
            n, nb = ws[0].numel(), bs[0].numel()
            ok = all(off[id(ws[i])] == off[id(ws[0])] + i * n for i in range(3)) and \
                all(off[id(bs[i])] == off[id(bs[0])] + i * nb for i in range(3)) and n % 4 == 0 and nb % 4 == 0
            if not ok:
                continue
            rows, cols = ws[0].shape
            ow, ob = off[id(ws[0])], off[id(bs[0])]
            w = self.flat.flat_param[ow:ow + 3 * n].view(3 * rows, cols).requires_grad_(True)
            b = self.flat.flat_param[ob:ob + 3 * nb].view(3 * nb).requires_grad_(True)
            w._focr_grad = self.flat.flat_grad[ow:ow + 3 * n].view(3 * rows, cols)
            b._focr_grad = self.flat.flat_grad[ob:ob + 3 * nb].view(3 * nb)
            m._packed_qkv = (w, b)

    def _attach_packed_gru(self):
        """TSRN GruBlock: the kernels want [W_ih | W_ih_reverse] (192 x 64), stacked W_hh (2 x 96 x 32) and the biases.
        The reference-named nn.GRU parameters are laid out pairwise-adjacent in the flat buffers (_qkv_adjacent_order),
        so the packed operands are views of the flat parameter buffer and their gradient targets views of the flat
        gradient: no torch.cat / stack per forward, no slice-gradient accumulation kernels per backward (4 cats +
        8 adds per GruBlock, 10 blocks per step)."""
        off = {id(p): o for p, o in zip(self.flat.params, self.flat.offsets)}
        for m in self.model.modules():
            g = getattr(m, "gru", None)
            if g is None or not hasattr(m, "_packed_gru") or not isinstance(g, torch.nn.GRU):
                continue
            pairs = [(g.weight_ih_l0, g.weight_ih_l0_reverse), (g.weight_hh_l0, g.weight_hh_l0_reverse),
                     (g.bias_ih_l0, g.bias_ih_l0_reverse), (g.bias_hh_l0, g.bias_hh_l0_reverse)]
            if any(id(a) not in off or id(b) not in off or a.numel() % 4 or off[id(b)] != off[id(a)] + a.numel()
                   for a, b in pairs):
                continue
            views = []
            for (a, _), shape in zip(pairs, ((2 * a.shape[0], a.shape[1]) if i == 0 else
                                             ((2,) + tuple(a.shape)) if i in (1, 3) else (2 * a.shape[0],)
                                             for i, (a, _) in enumerate(pairs))):
                o, n = off[id(a)], 2 * a.numel()
                v = self.flat.flat_param[o:o + n].view(shape).requires_grad_(True)
                v._focr_grad = self.flat.flat_grad[o:o + n].view(shape)
                views.append(v)
            m._packed_gru = (views[0], views[2], views[1], views[3])          # wih, bih, whh, bhh

    # ---- overlap plan ---------------------------------------------------------------------------
    def _offset_of(self, module):
        ids = {id(p): off for p, off in zip(self.flat.params, self.flat.offsets)}
        offs = [ids[id(p)] for p in module.parameters() if id(p) in ids]
        return min(offs) if offs else None

    def _plan_overlap(self, boundaries, tail):
        self.ranges = []                       # [(module name, (lo, hi))] in firing (backward) order
        mods = dict(self.model.named_children())
        main_end = self.flat.numel
        for t in tail:
            if t in mods and self._offset_of(mods[t]) is not None:
                main_end = min(main_end, self._offset_of(mods[t]))
        hi = main_end
        for name in reversed([b for b in boundaries if b in mods]):
            lo = self._offset_of(mods[name])
            if lo is None or lo >= hi:
                continue
            self.ranges.append((name, (lo, hi)))
            mods[name].register_forward_pre_hook(self._make_hook(len(self.ranges) - 1))
            hi = lo
        self.rest = [(0, hi)] + ([(main_end, self.flat.numel)] if main_end < self.flat.numel else [])
        self.comm_stream = torch.cuda.Stream() if (self.world > 1 and self.flat.flat_grad.is_cuda) else None

    def _make_hook(self, idx):
        def pre_hook(module, args):
            if self.world == 1 or not torch.is_grad_enabled() or not args[0].requires_grad:
                return None
            return (_Boundary.apply(args[0], lambda: self._launch(self.ranges[idx][1])),) + tuple(args[1:])
        return pre_hook

    def _launch(self, rng):
        lo, hi = rng
        if hi <= lo:
            return
        self._sent.append(rng)
        view = self.flat.flat_grad[lo:hi]
        if self.comm_stream is not None:
            # parked weight-gradient closures (FOCR_DEFER_SIDE=1) record their inputs-ready event on the CURRENT stream:
            # issue them here, on the compute stream, not inside the comm-stream context below (where they would order
            # themselves behind in-flight all-reduces)
            self.ctx.flush_side()
            self.ctx.flush_tail()                          # (a bucket with tail-parked gradients in it releases them now)
            ev = torch.cuda.Event()
            ev.record()                                    # gradients of this range are complete here ...
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self.ctx.join_side_stream(self.comm_stream)    # ... once the side-stream weight gradients are in too
                if self.native_comm:
                    from . import _lib
                    _lib.call("focr_allreduce_async", ctypes.c_void_p(view.data_ptr()), view.numel(), 0,
                              ctypes.c_void_p(self.comm_stream.cuda_stream))
                else:
                    self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def allreduce_grads(self):
        """All-reduce whatever the backward hooks have not sent yet, then wait for everything."""
        if self.world == 1:
            return
        timed = self.comm_timing and self.flat.flat_grad.is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()                  # backward and the side stream's weight gradients are complete here
        sent = sorted(self._sent)
        todo, pos = [], 0
        for lo, hi in sent + [(self.flat.numel, self.flat.numel)]:
            if lo > pos:
                todo.append((pos, lo))
            pos = max(pos, hi)
        for lo, hi in todo:                                   # few large messages (xGMI ring is per-link bound)
            step = max(4, (hi - lo + len(self.buckets) - 1) // len(self.buckets) // 4 * 4)
            for a in range(lo, hi, step):
                self._launch((a, min(hi, a + step)))
        for w in self._works:
            w.wait()
        if self.native_comm:
            from . import _lib
            tmo = os.environ.get("FOCR_COMM_TIMEOUT_MS")
            if tmo:     # watchdog (host sync): a lost peer becomes an error instead of a hang at the next step
                _lib.call("focr_comm_wait", ctypes.c_void_p(self.comm_stream.cuda_stream), int(tmo))
            else:
                _lib.call("focr_comm_async_error")
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if timed:
            ev1.record()                  # the compute stream continues (clip + Adam) only after every range has arrived
            self._comm_events.append((ev0, ev1))
        self._works, self._sent = [], []

    def rccl_ranks(self):
        """the rank count the MEASURED communicator reports: a sum all-reduce of 1 over the path the gradients take
        (torch.distributed's backend, or the library's own communicator with FOCR_COMM=native + ncclCommCount)"""
        if self.world == 1:
            return 1
        one = torch.ones(4, device=self.flat.flat_grad.device)
        if self.native_comm:
            from . import _lib
            _lib.call("focr_allreduce_async", ctypes.c_void_p(one.data_ptr()), 4, 0,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            n = int(round(one[0].item()))
            cnt = _lib.load().focr_comm_count()
            if cnt != n:
                raise RuntimeError("ncclCommCount says %d ranks, an all-reduce of 1 returned %d" % (cnt, n))
            return n
        dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.pg)
        return int(round(one[0].item()))

    def exposed_comm_ms(self):
        """mean over the timed steps (comm_timing = True) of the compute stream's wait for the gradient exchange"""
        if not self._comm_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._comm_events]
        self._comm_events = []
        return sum(ms) / len(ms)

    def _set_modes(self):
        """train mode (nn.Dropout slots in eval for parity runs).  Walking the 190 modules costs ~0.8 ms, so it is done
        when the mode flags can have changed: first call, or the root was switched to eval in between."""
        if self.model.training and getattr(self, "_modes_set", False):
            return
        self.model.train()
        if not self.dropout:
            for m in self.model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.eval()
        self._modes_set = True

    # ---- recorded step --------------------------------------------------------------------------------------------------
    def _replay_ok(self):
        from . import _lib
        return (self.replay and self.world == 1 and self.flat.flat_grad.is_cuda and _lib._timed is None
                and getattr(self.crit, "REPLAY_SAFE", False) and not torch.cuda.is_current_stream_capturing())

    def _rec_key(self, lr, hr, encoded):
        from . import _lib
        lib = _lib.load()
        # everything a recording bakes into its launches' scalar arguments / launch list: shapes, arithmetic and kernel
        # selection, the optimiser's hyper-parameters (a learning-rate change makes a new recording, never a silent no-op),
        # which parameters are trained
        return (tuple(lr.shape), tuple(hr.shape), encoded.key() if hasattr(encoded, "key") else encoded is not None,
                lib.focr_get_precision(),
                tuple(lib.focr_get_tuning(k) for k in range(6)), bool(self.dropout), bool(self.wgrad_side_stream),
                float(self.opt.lr), tuple(self.opt.betas), float(self.opt.eps), float(self.opt.max_norm),
                sum(1 for p in self.flat.params if p.requires_grad))

    @staticmethod
    def _fill(st, lr, hr, encoded):
        if lr is not st["lr"]:
            st["lr"].copy_(lr, non_blocking=True)
        if hr is not st["hr"]:
            st["hr"].copy_(hr, non_blocking=True)
        if encoded is not None and encoded is not st["enc"] and hasattr(encoded, "copy_into"):
            encoded.copy_into(st["enc"])                 # loss/padded_labels.py: one packed tensor
        elif encoded is not None and encoded is not st["enc"]:
            t, l, o = encoded[0], encoded[1], encoded[2]
            st["enc"][0][:t.numel()].copy_(t, non_blocking=True)
            st["enc"][1].copy_(l, non_blocking=True)
            st["enc"][2].copy_(o, non_blocking=True)

    def _record(self, key, lr, hr, encoded):
        """capture one step on static copies of the inputs and build its launch list (replay.record)"""
        import warnings
        from . import replay
        st = {"lr": torch.empty_like(lr), "hr": torch.empty_like(hr), "enc": None}
        if encoded is not None and hasattr(encoded, "static_like"):
            st["enc"] = encoded.static_like()
        elif encoded is not None:
            cap = max(int(encoded[0].numel()), lr.shape[0] * 32)
            st["enc"] = (torch.zeros(cap, device=lr.device, dtype=encoded[0].dtype), torch.zeros_like(encoded[1]),
                         torch.zeros_like(encoded[2]))
        self._fill(st, lr, hr, encoded)
        if self.ctx.side_stream_obj is None:
            self.ctx.side_stream_obj = torch.cuda.Stream()
        t_host = self.opt.t
        try:
            rec, out = replay.record(lambda: self._step(st["lr"], st["hr"], None, st["enc"]),
                                     lanes=[torch.cuda.current_stream(), self.ctx.side_stream_obj])
        except Exception as e:                                   # noqa: BLE001
            warnings.warn("fudanocr_amd: recording the training step failed (%s: %s); stepping eagerly from here on"
                          % (type(e).__name__, str(e)[:300]))
            self.replay = False
            self.ctx.deferred.clear()
            self.ctx.premasked.clear()
            return None
        finally:
            self.opt.t = t_host              # the captured step did not run
        st["rec"], st["out"] = rec, out
        # eval-mode BatchNorm caches of this model's layers are dropped by the train-mode forward in Python
        # (kernels._BatchNormAct); a replay runs no Python, so the engine drops them after every launch
        st["rvars"] = [id(m.running_var) for m in self.model.modules()
                       if isinstance(getattr(m, "running_var", None), torch.Tensor)]
        self._recs[key] = st
        return st

    def recorded_inputs(self, images_lr, images_hr, encoded=None):
        """the static input tensors of the recording that serves these shapes (or None): a loader that writes the next batch
        straight into them -- and passes them back in -- saves the per-step input copies"""
        st = self._recs.get(self._rec_key(images_lr, images_hr, encoded))
        return None if st is None else (st["lr"], st["hr"], st["enc"])

    def _call_recorded(self, images_lr, images_hr, label_strs, encoded):
        if encoded is None and label_strs is not None:
            # the criterion's labels as device tensors a recording can keep a static copy of: the focus losses pad them to a
            # capacity bucket (loss/padded_labels.py), the CTC criterion packs them (ctc_focus_loss.encode)
            if hasattr(self.crit, "encode_for_replay"):
                encoded = self.crit.encode_for_replay(label_strs, images_lr.device)
            elif getattr(self.crit, "recognizer", [None])[0] is not None:
                encoded = self.crit.encode(label_strs, images_lr.device)
        key = self._rec_key(images_lr, images_hr, encoded)
        st = self._recs.get(key)
        if st is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n <= self.REPLAY_WARMUP or len(self._recs) >= self.MAX_RECORDINGS:
                return self._step(images_lr, images_hr, label_strs, encoded)
            st = self._record(key, images_lr, images_hr, encoded)
            if st is None:
                return self._step(images_lr, images_hr, label_strs, encoded)
        if encoded is not None and encoded is not st["enc"] and (
                not encoded.fits(st["enc"]) if hasattr(encoded, "fits") else encoded[0].numel() > st["enc"][0].numel()):
            return self._step(images_lr, images_hr, label_strs, encoded)       # more label characters than the static buffer
        self._fill(st, images_lr, images_hr, encoded)
        st["rec"].launch()
        self.recorded = st["rec"]
        self.opt.t += 1
        K.bump_weight_epoch()                              # parameters changed behind autograd's version counters
        for i in st["rvars"]:
            K._EVAL_INVSTD.pop(i, None)
        return st["out"]

    def __call__(self, images_lr, images_hr, label_strs=None, encoded=None):
        """one optimisation step.  With replay active the returned tensors are STATIC (overwritten by the next step)."""
        self._set_modes()
        if self._replay_ok():
            return self._call_recorded(images_lr, images_hr, label_strs, encoded)
        return self._step(images_lr, images_hr, label_strs, encoded)

    def _step(self, images_lr, images_hr, label_strs=None, encoded=None):
        if self.state is not None:
            self.state.bind()
            self.state.advance()           # epoch + 1, t + 1, bias corrections of t: the first launch of the step
        try:
            return self._step_body(images_lr, images_hr, label_strs, encoded)
        finally:
            if self.state is not None:
                self.state.unbind()

    def _step_body(self, images_lr, images_hr, label_strs=None, encoded=None):
        self._set_modes()
        self.flat.zero_grad()
        self._works, self._sent = [], []
        on_gpu = self.flat.flat_grad.is_cuda
        c = self.ctx
        c.frags = self.frags if on_gpu else None
        flip_ev = None
        if on_gpu:
            self.frags.refresh()
            c.prefetch_masks()
            if self.flips.built and self.flips.n_live and self.wgrad_side_stream:
                # the flipped weights of the data-gradient GEMMs depend on the parameters only: re-flipped at the START of
                # the step on the side stream (idle during the forward pass) instead of between loss and backward on the
                # main stream (19 us + a dependent launch on the critical path); backward waits for the event
                # INVARIANT: nothing in the forward pass reads the FlipTable's buffers (c.flips is None until the
                # backward starts, asserted here) and nothing writes parameters between this point and the backward
                assert c.flips is None
                if c.side_stream_obj is None:
                    c.side_stream_obj = torch.cuda.Stream()
                side = c.side_stream_obj
                side.wait_stream(torch.cuda.current_stream())      # the previous step's optimiser and backward are behind it
                with torch.cuda.stream(side):
                    self.flips.refresh()
                    flip_ev = torch.cuda.Event()
                    flip_ev.record(side)
        try:
            with K.use_context(c):
                if _HR_SIDE and on_gpu and self.wgrad_side_stream and encoded is not None and hasattr(self.crit, "prefetch_hr"):
                    if c.side_stream_obj is None:
                        c.side_stream_obj = torch.cuda.Stream()
                    self.crit.prefetch_hr(images_hr, encoded, c.side_stream_obj)
                sr = self.model(images_lr)
                loss, mse, _, ctc = self.crit(sr, images_hr, label_strs, encoded)
            if on_gpu:
                if flip_ev is not None:
                    torch.cuda.current_stream().wait_event(flip_ev)
                else:
                    self.flips.refresh()
                c.flips = self.flips
            if on_gpu and self.wgrad_side_stream:
                # the zeroed flat gradient must be visible to the side stream before its kernels accumulate into it
                c.side_enabled = True
                c.side_stream().wait_stream(torch.cuda.current_stream())
            # (loss * 100).backward() of super_resolution.py:79-82, without its three scalar launches on the critical path
            # (the multiply, the seed fill, the multiply's backward): loss = mse [+ ctc], so the same gradient -- exactly
            # 100.0 -- is fed into the loss terms directly
            if self._g100 is None or self._g100.device != loss.device:
                self._g100 = torch.full((), 100.0, device=loss.device)
            # Only a criterion that DECLARES loss == mse + recognition term takes the shortcut (CTCFocusLoss.LOSS_IS_MSE_PLUS_REC);
            # TextFocusLoss / StrokeFocusLoss weight their terms (mse + 10 * attention + 0.0005 * recognition,
            # text_focus_loss.py:84-99) and go through (loss * 100).backward()
            terms = [t for t in (mse, ctc) if torch.is_tensor(t) and t.requires_grad]
            if (getattr(self.crit, "LOSS_IS_MSE_PLUS_REC", False) and torch.is_tensor(ctc) and loss.grad_fn is not None
                    and len(terms) == 2):
                torch.autograd.backward(terms, [self._g100.expand(t.shape) for t in terms])
            else:
                (loss * 100).backward()
            c.flush_side()             # weight gradients the last block parked (kernels.StepContext.defer_side)
            c.flush_tail()             # ... and anything still parked for the tail (no TPS warp in this model)
        finally:
            c.side_enabled = False
            c.flips = None
            c.frags = None
        if on_gpu:
            self.flips.build(self.flat.flat_grad.device)       # no-op after the first step
        c.check_deferred()                                 # every parked residual gradient was picked up
        if on_gpu:
            c.join_side_stream()                           # weight gradients complete before all-reduce / optimiser
        self.allreduce_grads()
        self.opt.step(self.world)
        K.bump_weight_epoch()                              # parameters changed behind autograd's version counters
        return {"loss": loss.detach(), "mse": mse.detach(),
                "ctc": ctc.detach() if torch.is_tensor(ctc) else None, "sr": sr.detach()}

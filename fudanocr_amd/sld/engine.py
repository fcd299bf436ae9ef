"""Optimisation step of the stroke-level-decomposition recognizer (reference train.py:63-77: zero_grad -> forward ->
CrossEntropyLoss -> backward -> Adadelta(lr=1.0, rho=0.9).step()), one process per GPU.

Same machinery as fudanocr_amd.engine.TrainStep: flat parameter / gradient buffers, per-engine StepContext (fragment
weights prepared once per step, weight gradients on a side stream), fused optimizer kernel, and -- replacing the
reference's nn.DataParallel (train.py:28) -- an RCCL all-reduce of the flat gradient.  This model's message is 287 MB
(71.7 M parameters), so unlike the SR nets the collective is bandwidth-relevant: the buffer goes out in
32 MB buckets, and the buckets behind an encoder stage are launched from autograd hooks as soon as backward has left
that stage (the flat buffer is in forward order, so everything behind a stage boundary is final), overlapping the
remaining backward kernels."""
import os

import torch
import torch.distributed as dist

from .. import kernels as K
from ..engine import FlatBuffers, _Boundary
from . import ops


class FusedAdadelta:
    def __init__(self, flat, lr=1.0, rho=0.9, eps=1e-6):
        self.flat, self.lr, self.rho, self.eps = flat, lr, rho, eps
        self.sq = torch.zeros_like(flat.flat_param)
        self.acc = torch.zeros_like(flat.flat_param)

    def step(self, world=1):
        ops.adadelta(self.flat.flat_param, self.flat.flat_grad, self.sq, self.acc, self.lr, self.rho, self.eps,
                     1.0 / world)


class SLDTrainStep:
    """model: sld.model.transformer.Transformer.  step(image, length, text_input, text_gt) -> {'loss', 'pred'}"""

    BUCKET = 8 << 20          # floats per all-reduce message (32 MB)

    REPLAY_WARMUP, MAX_RECORDINGS = 2, 2

    def __init__(self, model, lr=1.0, rho=0.9, process_group=None, wgrad_side_stream=True, dropout=True,
                 boundaries=("layer1", "layer2", "layer3", "layer4"), replay=None):
        self.model, self.dropout = model, dropout
        self.flat = FlatBuffers(list(model.parameters()))
        self.opt = FusedAdadelta(self.flat, lr, rho)
        self.ctx, self.frags, self.flips = K.StepContext(), K.FragTable(managed=True), K.FlipTable()
        self.pg, self.wgrad_side_stream = process_group, bool(wgrad_side_stream)
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        if self.world > 1:
            dist.broadcast(self.flat.flat_param, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                           group=process_group)
        self.comm_stream = torch.cuda.Stream() if (self.world > 1 and self.flat.flat_grad.is_cuda) else None
        self._works, self._sent_lo = [], self.flat.numel
        self.comm_timing, self._comm_events = False, []       # see engine.TrainStep.exposed_comm_ms
        # recorded step (engine.TrainStep has the long story): dropout epoch on the device, fixed per-site seeds, the step
        # captured once per input signature and re-issued from the library.  A signature is (batch, longest label, total
        # label length): real label batches vary, so at most MAX_RECORDINGS signatures are recorded (each holds a private
        # memory pool of the step's temporaries); everything else steps eagerly.
        on_gpu = self.flat.flat_grad.is_cuda
        self.state = K.StepState(self.flat.flat_grad.device, 0.5, 0.999) if on_gpu else None
        self.ctx.seed_base = int(torch.randint(0, 2 ** 62, (1,), device="cpu").item()) if on_gpu else None
        self.replay = (os.environ.get("FOCR_REPLAY", "1") != "0") if replay is None else bool(replay)
        self._recs, self._seen, self.recorded = {}, {}, None
        ids = {id(p): off for p, off in zip(self.flat.params, self.flat.offsets)}
        enc = model.encoder
        self._stage_lo = {}
        for name in boundaries:                       # flat offset where the stage AFTER this boundary's input begins
            mod = getattr(enc, name)
            offs = [ids[id(p)] for p in mod.parameters() if id(p) in ids]
            if offs:
                self._stage_lo[name] = min(offs)
                mod.register_forward_pre_hook(self._make_hook(name))

    def _make_hook(self, name):
        def pre_hook(module, args):
            if self.world == 1 or not torch.is_grad_enabled() or not args[0].requires_grad:
                return None
            # when the gradient arrives at this stage's INPUT, every parameter from this stage to the end is final
            return (_Boundary.apply(args[0], lambda: self._send_down_to(self._stage_lo[name])),) + tuple(args[1:])
        return pre_hook

    def _send_down_to(self, lo):
        """all-reduce flat_grad[lo : first offset already sent] in buckets"""
        hi = self._sent_lo
        if lo >= hi:
            return
        self._sent_lo = lo
        ev = None
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()
        for a in range(lo, hi, self.BUCKET):
            view = self.flat.flat_grad[a:min(hi, a + self.BUCKET)]
            if self.comm_stream is not None:
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_event(ev)
                    self.ctx.join_side_stream(self.comm_stream)
                    self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            else:
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    # ---- recorded step ------------------------------------------------------------------------------------------------
    def _replay_ok(self, length):
        from .. import _lib
        return (self.replay and self.world == 1 and self.flat.flat_grad.is_cuda and _lib._timed is None
                and getattr(length, "_focr_idx", None) is not None and not torch.cuda.is_current_stream_capturing())

    def _rec_key(self, image, text_input, text_gt):
        from .. import _lib
        lib = _lib.load()
        return (tuple(image.shape), tuple(text_input.shape), int(text_gt.numel()), lib.focr_get_precision(),
                tuple(lib.focr_get_tuning(k) for k in range(6)), bool(self.dropout), bool(self.wgrad_side_stream))

    @staticmethod
    def _fill(st, image, length, text_input, text_gt):
        if image is not st["image"]:
            st["image"].copy_(image, non_blocking=True)
            st["length"].copy_(length, non_blocking=True)
            st["text_input"].copy_(text_input, non_blocking=True)
            st["text_gt"].copy_(text_gt, non_blocking=True)
            st["length"]._focr_idx.copy_(length._focr_idx, non_blocking=True)

    def _record(self, key, image, length, text_input, text_gt):
        import warnings
        from .. import replay
        st = {"image": torch.empty_like(image), "length": torch.empty_like(length),
              "text_input": torch.empty_like(text_input), "text_gt": torch.empty_like(text_gt)}
        st["length"]._focr_idx = torch.empty_like(length._focr_idx)
        st["length"]._focr_host = None
        self._fill(st, image, length, text_input, text_gt)
        if self.ctx.side_stream_obj is None:
            self.ctx.side_stream_obj = torch.cuda.Stream()
        try:
            rec, out = replay.record(lambda: self._step(st["image"], st["length"], st["text_input"], st["text_gt"]),
                                     lanes=[torch.cuda.current_stream(), self.ctx.side_stream_obj])
        except Exception as e:                                   # noqa: BLE001
            warnings.warn("fudanocr_amd: recording the SLD step failed (%s: %s); stepping eagerly from here on"
                          % (type(e).__name__, str(e)[:300]))
            self.replay = False
            self.ctx.deferred.clear()
            self.ctx.premasked.clear()
            return None
        st["rec"], st["out"] = rec, out
        st["rvars"] = [id(m.running_var) for m in self.model.modules()
                       if isinstance(getattr(m, "running_var", None), torch.Tensor)]
        self._recs[key] = st
        return st

    def recorded_inputs(self, image, text_input, text_gt):
        st = self._recs.get(self._rec_key(image, text_input, text_gt))
        return None if st is None else (st["image"], st["length"], st["text_input"], st["text_gt"])

    def __call__(self, image, length, text_input, text_gt):
        if not self._replay_ok(length):
            return self._step(image, length, text_input, text_gt)
        key = self._rec_key(image, text_input, text_gt)
        st = self._recs.get(key)
        if st is None:
            n = self._seen[key] = self._seen.get(key, 0) + 1
            if n <= self.REPLAY_WARMUP or len(self._recs) >= self.MAX_RECORDINGS:
                return self._step(image, length, text_input, text_gt)
            st = self._record(key, image, length, text_input, text_gt)
            if st is None:
                return self._step(image, length, text_input, text_gt)
        self._fill(st, image, length, text_input, text_gt)
        st["rec"].launch()
        self.recorded = st["rec"]
        K.bump_weight_epoch()
        for i in st["rvars"]:
            K._EVAL_INVSTD.pop(i, None)
        return st["out"]

    def _step(self, image, length, text_input, text_gt):
        if self.state is not None:
            self.state.bind()
            self.state.advance()
        try:
            return self._step_body(image, length, text_input, text_gt)
        finally:
            if self.state is not None:
                self.state.unbind()

    def _step_body(self, image, length, text_input, text_gt):
        self.model.train()
        if not self.dropout:
            for m in self.model.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.eval()
        self.flat.zero_grad()
        self._works, self._sent_lo = [], self.flat.numel
        on_gpu = self.flat.flat_grad.is_cuda
        c = self.ctx
        c.new_step()                                   # dropout sites count from 0 in every step (fixed per-site seeds)
        c.frags = self.frags if on_gpu else None
        if on_gpu:
            self.frags.refresh()
        try:
            with K.use_context(c):
                result = self.model(image, length, text_input)
                loss = ops.cross_entropy(result["pred"], text_gt)
            if on_gpu:
                self.flips.refresh()
                c.flips = self.flips
            if on_gpu and self.wgrad_side_stream:
                c.side_enabled = True
                c.side_stream().wait_stream(torch.cuda.current_stream())
            loss.backward()
        finally:
            c.side_enabled, c.flips, c.frags = False, None, None
        if on_gpu:
            self.flips.build(self.flat.flat_grad.device)
            c.join_side_stream()
        if self.world > 1:
            timed = self.comm_timing and on_gpu
            if timed:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            self._send_down_to(0)
            for w in self._works:
                w.wait()
            if timed:
                ev1.record()
                self._comm_events.append((ev0, ev1))
        self.opt.step(self.world)
        K.bump_weight_epoch()
        return {"loss": loss.detach(), "pred": result["pred"].detach()}

    def rccl_ranks(self):
        """sum all-reduce of 1 over the communicator the gradient buckets use"""
        if self.world == 1:
            return 1
        one = torch.ones(4, device=self.flat.flat_grad.device)
        dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.pg)
        return int(round(one[0].item()))

    def exposed_comm_ms(self):
        if not self._comm_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._comm_events]
        self._comm_events = []
        return sum(ms) / len(ms)

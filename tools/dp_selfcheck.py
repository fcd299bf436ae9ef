#!/usr/bin/env python3
"""Data-parallel self-check for the first real multi-GPU run.  Launch exactly like the bench:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 \
         tools/dp_selfcheck.py [--config c3|c5] [--steps 3] [--batch N]

Every rank: binds its GPU, joins RCCL (backend nccl; FOCR_BENCH_BACKEND=gloo for ranks sharing one GPU), runs `steps`
optimisation steps of the configuration and checks
  * the world size RCCL reports == WORLD_SIZE, every rank sits on its own device,
  * the gradient buckets behind the stage boundaries were launched DURING backward (overlap path taken),
  * the all-reduced flat gradient equals the sum of the ranks' local gradients (one extra all_gather of a checksum
    vector per rank: sum / abs-sum / 3 probes -- no 12.8 MB gathers),
  * (c3) SURVEY 8(e)'s stronger identity: the reduced gradient equals the gradient ONE rank computes for the concatenated
    global batch with per-shard BatchNorm statistics, i.e. the sum over shards of the single-shard gradients, each
    recomputed locally from the shard's seed (max abs difference relative to the gradient's max, dropout off),
  * parameters are bit-identical across ranks after the steps,
and prints per-rank ms/step plus the shard loader's images/s on a synthetic uint8 shard (pinned staging + async H2D +
device transform, no training step attached); rank 0 prints one JSON line with the verdict.  Exit code != 0 on any
failure."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def loader_throughput(dev, rank, world, n=4096, batch=128, epochs=3):
    """images/s of dataset/shards.py ShardLoader alone (gather thread -> pinned slot -> async H2D -> device ToTensor) on a
    synthetic TextZoom-shaped uint8 shard in a temporary directory; the first epoch warms the page cache"""
    import json as _json
    import tempfile
    import numpy as np
    from fudanocr_amd.dataset.shards import ShardDataset, ShardLoader
    with tempfile.TemporaryDirectory() as d:
        rs = np.random.RandomState(7)
        rs.randint(0, 256, (n, 32, 128, 3), dtype=np.uint8).tofile(os.path.join(d, "hr.u8"))
        rs.randint(0, 256, (n, 16, 64, 3), dtype=np.uint8).tofile(os.path.join(d, "lr.u8"))
        open(os.path.join(d, "labels.txt"), "w").write("\n".join("w%d" % i for i in range(n)) + "\n")
        _json.dump({"n": n, "hr": [32, 128, 3], "lr": [16, 64, 3], "format": 1}, open(os.path.join(d, "meta.json"), "w"))
        ld = ShardLoader(ShardDataset(d), batch, dev, rank=rank, world=world)
        for _ in ld:
            pass
        torch.cuda.synchronize()
        t0, seen = time.perf_counter(), 0
        for _ in range(epochs):
            for hr, lr, _ in ld:
                seen += hr.shape[0]
        torch.cuda.synchronize()
        return seen / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c5"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-loader", action="store_true", help="skip the shard-loader throughput measurement")
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FOCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1:
        dist.init_process_group("nccl", device_id=dev) if backend == "nccl" else dist.init_process_group(backend)
    from fudanocr_amd import _lib
    _lib.load()
    fails = []
    if world > 1 and dist.get_world_size() != world:
        fails.append("world size %d != WORLD_SIZE %d" % (dist.get_world_size(), world))
    if a.config == "c5":
        from fudanocr_amd.sld import util as U
        from fudanocr_amd.sld.engine import SLDTrainStep
        from fudanocr_amd.sld.model.transformer import Transformer
        from fudanocr_amd.sld.synth import make_sld_batch
        from fudanocr_amd.utils.weight_fill import fill_module_
        net = fill_module_(Transformer("stroke")).to(dev)
        eng = SLDTrainStep(net, dropout=False)
        image, labels = make_sld_batch(a.batch or 8, 1234 + rank)
        length, ti, tg, _ = U.converter("stroke", labels, device=dev, strokes=True)
        image = image.to(dev)
        run = lambda: eng(image, length, ti, tg)                      # noqa: E731
        sent = lambda: eng._sent_lo                                   # noqa: E731
    else:
        from fudanocr_amd.engine import TrainStep
        from fudanocr_amd.smoke import build_models
        from fudanocr_amd.utils.synth import make_batch
        net, rec, crit = build_models(dev, "tbsrn")
        eng = TrainStep(net, crit, dropout=False)
        lr, hr, labels = make_batch(a.batch or 16, 1234 + rank)
        lr, hr = lr.to(dev), hr.to(dev)
        enc = crit.encode(labels, dev)
        run = lambda: eng(lr, hr, encoded=enc)                        # noqa: E731
        sent = None
    # ---- reference for the stronger identity (c3): every rank recomputes the single-shard gradient of EVERY shard with
    # the shared initial weights (world-1 code path of the same engine, optimiser stubbed out) and sums them
    expected = None
    if a.config == "c3" and world > 1:
        keep_world, keep_step = eng.world, eng.opt.step
        grads = []
        eng.world = 1
        eng.opt.step = lambda *_a, **_k: grads.append(eng.flat.flat_grad.detach().clone())
        for r in range(world):
            lr_r, hr_r, labels_r = make_batch(a.batch or 16, 1234 + r)
            eng(lr_r.to(dev), hr_r.to(dev), encoded=crit.encode(labels_r, dev))
        eng.world, eng.opt.step = keep_world, keep_step
        expected = torch.stack(grads).sum(0)
        eng.ctx._mask_cursor = 0
    # ---- step 1 with a gradient audit: snapshot the LOCAL gradient through a hook on the optimiser
    audit = {}
    orig_step = eng.opt.step

    def audited(worldsize=1):
        audit["reduced"] = eng.flat.flat_grad.detach().clone()
        return orig_step(worldsize)
    eng.opt.step = audited
    if world > 1:
        # local gradient of this rank for the same weights: a world-1 engine on a private group is not needed -- the
        # all-reduce is linear, so sum over ranks of (checksums of local gradients) must equal the reduced checksums.
        orig_ar = dist.all_reduce
        local_parts = []

        def spy(t, *args, **kw):
            local_parts.append((t.data_ptr(), t.detach().clone()))
            return orig_ar(t, *args, **kw)
        dist.all_reduce = spy
    t0 = time.perf_counter()
    out = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.all_reduce = orig_ar
        base = eng.flat.flat_grad.data_ptr()
        localg = torch.zeros_like(eng.flat.flat_grad)
        covered = 0
        for ptr, t in local_parts:
            off = (ptr - base) // 4
            if 0 <= off < localg.numel():
                localg[off:off + t.numel()] = t
                covered += t.numel()
        if covered != localg.numel():
            fails.append("all-reduce covered %d of %d gradient elements" % (covered, localg.numel()))
        probe = torch.stack([localg.double().sum(), localg.double().abs().sum(), localg[::997].double().sum(),
                             localg[1::4099].double().sum()])
        dist.all_reduce(probe)
        red = audit["reduced"]
        mine = torch.stack([red.double().sum(), red.double().abs().sum(), red[::997].double().sum(),
                            red[1::4099].double().sum()])
        # abs-sum is not linear: compare it loosely (it only has to be of the right magnitude), the sums tightly
        for i in (0, 2, 3):
            if abs(mine[i] - probe[i]) > 1e-4 * (abs(probe[i]) + probe[1] * 1e-6):
                fails.append("reduced gradient checksum %d: %.6e vs sum of locals %.6e" % (i, mine[i], probe[i]))
        if expected is not None:
            err = (red - expected).abs().max().item() / (expected.abs().max().item() + 1e-30)
            if not err <= 1e-4:          # fp32 atomics order in the backward reductions: ~1e-6; a wrong shard / scale: O(1)
                fails.append("reduced gradient != sum of single-shard gradients (per-shard BN): rel err %.3e" % err)
            audit["identity_err"] = err
        if a.config == "c3" and len(eng._sent) == 0 and len(local_parts) < 2:
            fails.append("no gradient bucket was launched during backward")
    eng.opt.step = orig_step
    times = []
    for _ in range(max(0, a.steps - 1)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = run()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    if not torch.isfinite(out["loss"]).item():
        fails.append("non-finite loss")
    if world > 1:
        mine = eng.flat.flat_param.detach().clone()
        other = mine.clone()
        dist.broadcast(other, src=0)
        if not torch.equal(mine, other):
            fails.append("parameters differ from rank 0 by %.3e" % (mine - other).abs().max().item())
        devs = [None] * world
        dist.all_gather_object(devs, (rank, local, torch.cuda.get_device_name(local)))
        if backend == "nccl" and len({d[1] for d in devs}) != world:
            fails.append("ranks share devices: %s" % devs)
    loader_ips = None
    if not a.no_loader:
        loader_ips = loader_throughput(dev, rank, world)
    print("rank %d dev %d: ms/step %s loss %.5f identity_err %s loader %s img/s%s" % (
        rank, local, ["%.2f" % t for t in times], out["loss"].item(),
        "%.2e" % audit["identity_err"] if "identity_err" in audit else "n/a",
        "%.0f" % loader_ips if loader_ips else "n/a", " FAIL " + "; ".join(fails) if fails else ""), flush=True)
    allf = [None] * world
    if world > 1:
        dist.all_gather_object(allf, fails)
    else:
        allf = [fails]
    if rank == 0:
        print(json.dumps({"dp_selfcheck": "ok" if not any(allf) else "FAILED", "world": world, "config": a.config,
                          "backend": backend, "ms_per_step_rank0": times, "failures": allf,
                          "identity_err_rank0": audit.get("identity_err"), "loader_img_per_s_rank0": loader_ips}))
    if world > 1:
        dist.destroy_process_group()
    sys.exit(1 if any(allf) else 0)


if __name__ == "__main__":
    main()

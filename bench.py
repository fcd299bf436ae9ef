#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the SR + CTC optimisation step on synthetic
16x64 -> 32x128 crops (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = forward (TBSRN, STN on, dropout on) -> MSE + frozen-CRNN CTC -> (loss*100).backward()
-> [RCCL all-reduce of the flat gradient buffer] -> clip 0.25 -> Adam, on a device-resident
synthetic batch of 128 images per GPU (BASELINE configs[2]/[3]; weak scaling).  Prints ONE JSON
line on rank 0 with `roofline` (dominant kernel, measured with on-stream events in the timed
region) and `cpu_baseline` (the CPU oracle timed on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = vector f32 peak
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA peak; the bf16x3 split issues 3 MFMA flops per algorithmic flop
FLOP_PER_IMG = 18.131e9           # TBSRN fwd+bwd 15.311 + frozen CRNN 2.820


def cpu_baseline(batch=4, budget_s=25.0):
    """Oracle (CPU restatement of the reference maths, kind 'port') on the host cores.
    Bounded sample: batch 4, as many timed steps as fit in `budget_s` seconds (at least one)."""
    from fudanocr_amd.utils.synth import make_batch
    from fudanocr_amd.utils.weight_fill import fill_dict_
    from oracle import sr_oracle as O
    cores = max(1, min(os.cpu_count() or 1, 32))      # beyond ~32 threads these small ops only slow down
    torch.set_num_threads(cores)
    P = O.make_params(O.schema_sr("tbsrn"))
    fill_dict_({k: v.data for k, v in P.items()})
    C = O.make_params(O.schema_crnn(), requires_grad=False)
    fill_dict_(C)
    opt = O.AdamState([v for v in P.values() if v.requires_grad])
    lr, hr, labels = make_batch(batch, 1234)
    tgt, tlen = O.encode_labels(labels)
    t0 = time.perf_counter()
    O.train_step(P, opt, "tbsrn", lr, hr, C, tgt, tlen, dropout_p=0.1)          # warm-up (also a size probe)
    warm = time.perf_counter() - t0
    steps, t0 = 0, time.perf_counter()
    while True:
        O.train_step(P, opt, "tbsrn", lr, hr, C, tgt, tlen, dropout_p=0.1)
        steps += 1
        dt = time.perf_counter() - t0
        if steps >= 5 or dt + warm + dt / steps > budget_s:
            break
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d timed steps of batch %d (TBSRN+CRNN-CTC step, fp32, torch CPU oracle, %d threads)"
                      % (steps, batch, cores)}


def cpu_baseline_guarded(timeout_s=150):
    """Run the CPU leg in a child process so that a pathological host (thread oversubscription,
    page-in stalls) can never hang the benchmark; returns a value-less record on timeout."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], timeout=timeout_s,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:                                   # noqa: BLE001
        return {"value": None, "unit": "images/sec", "cores": os.cpu_count(), "kind": "port",
                "sample": "CPU oracle leg did not finish within %ds (%s)" % (timeout_s, type(e).__name__)}


def main():
    if "--cpu-baseline-only" in sys.argv:
        print(json.dumps(cpu_baseline()))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--arch", default="tbsrn")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "bf16x3-allsplit", "bf16x3-dgrad16", "fp32"],
                    help="contraction arithmetic: split-bf16 MFMA with single-bf16 gradient accumulations in the "
                         "attention backward (library default, focr_set_precision(2)); the same with split "
                         "products everywhere (mode 1); or exact fp32 MFMA (mode 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # FOCR_BENCH_BACKEND=gloo lets several ranks share one GPU (functional check of the multi-process path
    # on a 1-GPU box; RCCL refuses two ranks on one device).  The measured configuration is always nccl = RCCL.
    backend = os.environ.get("FOCR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from fudanocr_amd import _lib
    from fudanocr_amd.engine import TrainStep
    from fudanocr_amd.smoke import build_models
    from fudanocr_amd.utils.synth import make_batch
    _lib.load()
    _lib.set_precision({"bf16x3": 2, "bf16x3-allsplit": 1, "bf16x3-dgrad16": 3, "fp32": 0}[args.precision])
    net, rec, crit = build_models(dev, args.arch)
    step = TrainStep(net, crit, dropout=True, wgrad_side_stream=os.environ.get("FOCR_WGRAD_SIDE", "1") != "0")
    lr, hr, labels = make_batch(args.batch, 1234 + rank)
    lr, hr = lr.to(dev), hr.to(dev)
    enc = crit.encode(labels, dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(lr, hr, encoded=enc)
    timed = ["focr_attention_fwd", "focr_attention_bwd"]          # 10 launches per step
    conv_steps = min(2, args.steps)      # the ~107 conv launches/step are event-timed in the last steps only (each
                                         # event pair costs host time: keeps the perturbation of `value` < 0.5 %)
    sync()
    _lib.start_timing(timed)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i == args.steps - conv_steps:
            _lib.add_timing(["focr_conv2d_fwd", "focr_conv3x3_frag_fwd"])
        out = step(lr, hr, encoded=enc)
    sync()
    dt = time.perf_counter() - t0
    kt = _lib.stop_timing_with_args()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss = out["loss"].item()
    if rank == 0:
        imgs = args.batch * world * args.steps
        value = imgs / dt
        bx3 = args.precision != "fp32"
        # ---- roofline of the dominant kernel: conv_fwd_bx3_kernel (implicit-GEMM conv / linear, forward AND
        # data-gradient launches; 28 % of the step in profiles/r01j).  Every C-ABI call = one kernel launch, timed with
        # events on its stream inside the timed region; algorithmic bytes = each input / weight / output (and
        # residual) element once, algorithmic flops = 2*M*K*N (DESIGN.md "Measurement").
        nbytes = nflops = ms = 0.0
        ncalls = 0
        for t_ms, a in kt["focr_conv2d_fwd"]:
            n, h, w, cin, cout, kh, kw, ph, pw = (int(v) for v in a[5:14])
            if bx3 and cin % 32:
                continue                              # tiny-Cin first layers run on the fp32 kernel
            oh, ow = h + 2 * ph - kh + 1, w + 2 * pw - kw + 1
            m = n * oh * ow
            has_res = bool(getattr(a[3], "value", a[3]))
            nbytes += 4.0 * (n * h * w * cin + cout * kh * kw * cin + m * cout * (2 if has_res else 1))
            nflops += 2.0 * m * cout * kh * kw * cin
            ms += t_ms
            ncalls += 1
        conv_gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        conv_tf = nflops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_BF16_MFMA_TFLOPS if bx3 else PEAK_F32_MFMA_TFLOPS
        fwd = [t for t, _ in kt["focr_attention_fwd"]]
        bwd = [t for t, _ in kt["focr_attention_bwd"]]
        fwd_ms = sum(fwd) / max(1, len(fwd))
        bwd_ms = sum(bwd) / max(1, len(bwd))
        flops_launch = 4.0 * args.batch * 4 * 1024 * 1024 * 32
        ach = flops_launch / (fwd_ms * 1e-3) / 1e12 if fwd_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes (profiles/README.md, r01p): (2*FETCH_SIZE + WRITE_SIZE) KB averaged
        # over the 107 focr_conv2d_fwd launches of a step (conv_fwd_bx3<1|2> + linear_stream kernels), measured at
        # per-GPU batch 128 only
        traffic = None
        if args.batch == 128 and bx3:
            traffic = 167.6e6
        res = {
            "metric": "training images/sec (16x64->32x128 SR+CTC step)", "value": round(value, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3" if bx3 else "f32", "data": "synthetic",
            "config": {"workload": "%s + frozen CRNN-CTC train step (BASELINE configs[2]%s), STN on, "
                                   "dropout on, 16x64->32x128" % (args.arch.upper(), "" if args.arch == "tbsrn"
                                                                  else "; architecture variant"),
                       "per_gpu_batch": args.batch,
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "arch": args.arch,
                       "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "arithmetic": ("split-bf16 MFMA (hi/lo operands, 3 products, fp32 accumulate)" +
                                      ("; dV/dK/dQ accumulations of the attention backward in single bf16 products"
                                       if args.precision == "bf16x3" else "")) if bx3
                       else "exact fp32 MFMA"},
            # the bounding roofline of this path is the dense-contraction (MFMA) one; `achieved` is the
            # ALGORITHMIC flop rate of the dominant kernel's launches (bf16x3 executes 3 MFMA flops per algorithmic
            # flop: executed_frac).  hbm_view: the same launches against HBM -- the roof that is actually closer
            # for these layers.
            "roofline": {"bound": "mfma",
                         "kernel": ("conv_fwd_bx3_kernel / linear_stream_bx3_kernel" if bx3 else "conv_fwd_kernel") +
                                   " (focr_conv2d_fwd: implicit-GEMM conv + streaming linear, forward and "
                                   "data-gradient launches)",
                         "achieved": round(conv_tf, 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(conv_tf / peak, 4), "traffic": traffic,
                         "executed_frac": round((3 if bx3 else 1) * conv_tf / peak, 4),
                         "hbm_view": {"achieved": round(conv_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                      "frac": round(conv_gbs / PEAK_HBM_GBS, 4)},
                         "launches_per_step": ncalls // max(1, conv_steps),
                         "avg_launch_ms": round(ms / max(1, ncalls), 4),
                         "algorithmic_bytes_per_launch": round(nbytes / max(1, ncalls), 0),
                         "algorithmic_flops_per_launch": round(nflops / max(1, ncalls), 0),
                         "also": [{"kernel": ("attn_fwd_bx3_kernel" if bx3 else "attn_fwd_kernel") +
                                             " (fused QK^T-softmax-dropout-PV, incl. keep-bit pre-pass)",
                                   "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                                   "frac": round(ach / peak, 4), "avg_launch_ms": round(fwd_ms, 4),
                                   "executed_frac": round((3 if bx3 else 1) * ach / peak, 4)},
                                  {"kernel": "attention backward (prep + dK/dV + dQ launches)", "bound": "mfma",
                                   "achieved": round(2.5 * flops_launch / (bwd_ms * 1e-3) / 1e12, 2) if bwd_ms else 0.0,
                                   "peak": peak, "unit": "TFLOP/s", "avg_launch_ms": round(bwd_ms, 4)}],
                         "step_algorithmic_tflops": round(value * FLOP_PER_IMG / world / 1e12, 2)},
            "final_loss": round(loss, 5),
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline_guarded()
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

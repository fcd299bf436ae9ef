"""Data-parallel engine on the GPU with two ranks sharing the one device over gloo (RCCL refuses two ranks on
one GPU, and the test box has one): the real TBSRN + CRNN-CTC step with the boundary hooks, the side stream,
the bucketed all-reduce and the fused clip+Adam.  Checks, per rank:
  * the boundary buckets are launched DURING backward (overlap path taken);
  * the all-reduced flat gradient equals the sum of both ranks' locally computed gradients;
  * after 2 optimizer steps the parameters of both ranks are identical;
  * bench.py's multi-process path prints its one JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from fudanocr_amd import _lib
from fudanocr_amd.engine import TrainStep
from fudanocr_amd.smoke import build_models
from fudanocr_amd.utils.synth import make_batch
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
_lib.load()
solo = dist.new_group([0]), dist.new_group([1])          # every rank creates both groups
net, rec, crit = build_models(dev, "tbsrn")
step = TrainStep(net, crit, dropout=False)
assert step.world == 2 and step.comm_stream is not None
lr, hr, labels = make_batch(4, 77 + rank)
lr, hr = lr.to(dev), hr.to(dev)
enc = crit.encode(labels, dev)

def local_backward(engine, model):
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.eval()
    engine.flat.zero_grad()
    engine._works, engine._sent = [], []
    sr = model(lr)
    loss = engine.crit(sr, hr, None, enc)[0]
    (loss * 100).backward()

# data-parallel backward
local_backward(step, net)
assert len(step._sent) >= 2, step._sent                    # boundary buckets went out during backward
step.allreduce_grads()
torch.cuda.synchronize()
dp = step.flat.flat_grad.clone()
# reference: same weights, world-1 engine, explicit sum over ranks
net1, _, crit1 = build_models(dev, "tbsrn")
net1.load_state_dict(net.state_dict())
one = TrainStep(net1, crit1, dropout=False, process_group=solo[rank])
assert one.world == 1
local_backward(one, net1)
ref = one.flat.flat_grad.clone()
dist.all_reduce(ref)
err = (dp - ref).abs().max().item() / ref.abs().max().item()
assert err < 1e-4, err
# two full steps, then parameters must agree bit-for-bit across ranks (same reduced gradients, same update)
for _ in range(2):
    out = step(lr, hr, encoded=enc)
assert torch.isfinite(out["loss"]).item()
mine = step.flat.flat_param.detach().clone()
other = mine.clone()
dist.broadcast(other, src=0)
assert torch.equal(mine, other), (mine - other).abs().max().item()
dist.destroy_process_group()
print("rank", rank, "ok", err)
"""


def _run_two(cmds_env, timeout=600):
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for cmd, env in cmds_env]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=timeout)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    return procs, outs


def test_dp_engine_two_ranks_one_gpu(tmp_path):
    script = tmp_path / "dp_gpu_worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs, outs = _run_two([([sys.executable, str(script)], dict(env, RANK=str(r))) for r in range(2)])
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])


def test_bench_multiprocess_path(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2",
               FOCR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "8"]
    procs, outs = _run_two([(cmd, dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    line = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(line) == 1 and not [l for l in outs[1].splitlines() if l.startswith("{")]
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 16 and res["value"] > 0
    assert res["scaling"] == "weak" and "cpu_baseline" not in res


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher environment (the shape of the driver's N > 1 bench command when it is
    not wrapped in torch.distributed.run): bench.py re-launches itself as two ranks and prints ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR",
                                                              "MASTER_PORT")}
    env["FOCR_BENCH_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "8"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout[-2000:]
    res = json.loads(line[0])
    assert res["n_gpus"] == 2 and res["config"]["collective_backend"] == "gloo" and res["value"] > 0
    assert res["config"]["global_batch"] == 16
    # scaling-run evidence carried by every N > 1 line: the rank count an all-reduce of 1 over the measured communicator
    # returns, the collective library's version (None on gloo) and the event-timed exposed communication per step
    assert res["rccl_ranks"] == 2 and "rccl_version" in res and res["comm_path"].startswith("torch.distributed")
    assert res["exposed_comm_ms"] is not None and res["exposed_comm_ms"] >= 0.0


MAIN_WORKER = r"""
import os, sys, yaml, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.chdir(%r)
from fudanocr_amd import main as M
from fudanocr_amd.utils.util import AttrDict
cfg = AttrDict(yaml.load(open(os.path.join(os.path.dirname(M.__file__), "config", "super_resolution.yaml")),
                         Loader=yaml.Loader))
cfg.TRAIN.iters_per_epoch, cfg.TRAIN.displayInterval, cfg.TRAIN.saveInterval = 3, 1, 2
captured = {}
from fudanocr_amd.interfaces import base as B
orig = B.TextBase.optimizer_init
def spy(self, model, crit):
    captured["step"] = orig(self, model, crit)
    return captured["step"]
B.TextBase.optimizer_init = spy
res = M.main(cfg, M.parse(["--arch", "tbsrn", "--STN", "--exp_name", "dp", "--batch_size", "4"]))
step = captured["step"]
assert dist.is_initialized() and step.world == 2, "main.py did not join the process group"
mine = step.flat.flat_param.detach().clone()
other = mine.clone()
dist.broadcast(other, src=0)
assert torch.equal(mine, other), "replicas diverged"
rank = dist.get_rank()
print("rank", rank, "ok; checkpoint written:", os.path.exists("checkpoint/dp/model_best.pth"))
dist.destroy_process_group()
"""


def test_main_py_under_two_ranks(tmp_path):
    """`python -m torch.distributed.run ... -m fudanocr_amd.main` must train ONE data-parallel job, not N
    independent replicas: TextBase joins the process group and binds the device, ranks draw different shards, only
    rank 0 wipes / writes the checkpoint directory, parameters stay bit-identical."""
    script = tmp_path / "main_dp_worker.py"
    script.write_text(MAIN_WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2", FOCR_DIST_BACKEND="gloo")
    procs, outs = _run_two([([sys.executable, str(script)], dict(env, RANK=str(r), LOCAL_RANK=str(r)))
                            for r in range(2)])
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    assert os.path.exists(tmp_path / "checkpoint" / "dp" / "model_best.pth")
    assert "log.txt" in os.listdir(tmp_path / "checkpoint" / "dp")


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_dp_selfcheck_tool(config):
    """tools/dp_selfcheck.py (what the 8-GPU driver run can invoke) under 2 ranks on one GPU (gloo): checks world
    size, overlap launches, reduced gradient == sum of locals (checksums), bit-identical parameters"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541" if config == "c3" else "29543", WORLD_SIZE="2",
               FOCR_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "dp_selfcheck.py"), "--config", config, "--steps", "3",
           "--batch", "4"]
    procs, outs = _run_two([(cmd, dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    line = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert json.loads(line[-1])["dp_selfcheck"] == "ok"


@pytest.mark.gpu
def test_c_abi_rccl_communicator_single_rank():
    """include/focr.h focr_comm_*: the library's own RCCL communicator (bound with dlopen).  One GPU can host one rank
    only (RCCL refuses two ranks on a device), so this exercises the full call sequence at world size 1: unique id ->
    init -> in-place sum all-reduce on a side stream (identity at one rank) -> destroy; plus the error paths."""
    import ctypes
    from fudanocr_amd import _lib
    lib = _lib.load()
    assert lib.focr_comm_nranks() == 0
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    with pytest.raises(RuntimeError):                       # no communicator yet
        _lib.call("focr_allreduce_async", ctypes.c_void_p(x.data_ptr()), x.numel(), 0, ctypes.c_void_p(0))
    ident = ctypes.create_string_buffer(128)
    _lib.call("focr_comm_unique_id", ident)
    assert any(ident.raw)
    _lib.call("focr_comm_init", 0, 1, ident)
    try:
        assert lib.focr_comm_nranks() == 1
        with pytest.raises(RuntimeError):                   # a second communicator is refused
            _lib.call("focr_comm_init", 0, 1, ident)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        ref = x.clone()
        for _ in range(3):
            _lib.call("focr_allreduce_async", ctypes.c_void_p(x.data_ptr()), x.numel(), 0,
                      ctypes.c_void_p(side.cuda_stream))
        _lib.call("focr_comm_async_error")                   # healthy communicator: no asynchronous error
        _lib.call("focr_comm_wait", ctypes.c_void_p(side.cuda_stream), 20000)   # watchdog: drains well inside the limit
        assert lib.focr_comm_nranks() == 1                   # ... and did not abort the communicator
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        with pytest.raises(RuntimeError):                   # only fp32 gradient buffers
            _lib.call("focr_allreduce_async", ctypes.c_void_p(x.data_ptr()), x.numel(), 1, ctypes.c_void_p(0))
    finally:
        _lib.call("focr_comm_destroy")
    assert lib.focr_comm_nranks() == 0

#!/bin/bash
# N > 1 path of the final code on one device (2 ranks, gloo): dp_selfcheck (c3) and the bench launched as the driver does
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FOCR_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 tools/dp_selfcheck.py --config c3 --steps 3 > gpurun_out/c31_dp.log 2>&1
grep -E "dp_selfcheck|rank [01] dev" gpurun_out/c31_dp.log | cut -c1-300
FOCR_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29634 bench.py --gpus 2 --steps 10 --warmup 5 > gpurun_out/c31_bench2.log 2>&1
grep "^{" gpurun_out/c31_bench2.log | cut -c1-330

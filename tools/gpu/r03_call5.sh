#!/bin/bash
# round 3, call 5: profile artefacts of the round-3 kernel set (kernel trace + FETCH / WRITE / MFMA PMC passes), host
# profile, side-stream A/B, 2-rank dp_selfcheck (gloo, one device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round.sh r03a > gpurun_out/c5_profile.log 2>&1
timeout 300 python tools/dev/host_profile.py > gpurun_out/c5_host.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/c5_b_side1.log 2>&1
FOCR_WGRAD_SIDE=0 timeout 200 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/c5_b_side0.log 2>&1
timeout 200 python bench.py --no-cpu-baseline --steps 40 > gpurun_out/c5_b_side1b.log 2>&1
FOCR_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 tools/dp_selfcheck.py --config c3 --steps 3 > gpurun_out/c5_dp.log 2>&1
tail -3 gpurun_out/c5_dp.log | cut -c1-400
for f in side1 side0 side1b; do python - <<PY
import json
for l in open('gpurun_out/c5_b_$f.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$f', d['value'], d['ms_per_step'])
PY
done
head -14 gpurun_out/c5_host.log | tail -8

#!/bin/bash
# round 5, call 25: host side of the step: enqueue time at a small batch, cProfile, step time against batch (host floor)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 python tools/dev/host_time.py 2>&1 | tail -3 | tee gpurun_out/r05_c25_host_time.txt
timeout 300 python tools/dev/host_profile.py 2>&1 | head -50 > gpurun_out/r05_c25_host_profile.txt; head -45 gpurun_out/r05_c25_host_profile.txt
for b in 8 32 64 128; do timeout 300 python bench.py --steps 30 --warmup 10 --batch $b --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('B=$b', d['ms_per_step'], d['value'])"; done | tee gpurun_out/r05_c25_batch_sweep.txt

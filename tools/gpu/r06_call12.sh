#!/bin/bash
# round 6, call 12: GRU scans as loader / compute wave pairs: identity test, fp64 test, micro-benchmark A/B, c1 step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gru" 2>&1 | tail -8
GRU_TUNE=0,1,2 timeout 200 python tools/dev/gru_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c12_gru_bench.txt
for T in 2 1 2 1; do timeout 600 python bench.py --config c1 --steps 40 --warmup 10 --no-cpu-baseline --tuning 5=$T 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c1 gru_loader=$T', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r06_c12_c1.txt

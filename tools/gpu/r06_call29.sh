#!/bin/bash
# round 6, call 29: WIDE row-streaming weight gradient standalone (correct workspace query), tfl / sfl / c3 / c1 regression timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for X in 1 0; do echo "== FOCR_C3W_WIDE=$X"; FOCR_C3W_WIDE=$X timeout 300 python tools/dev/wgrad_wide_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_c29_c3w_wide.txt
for C in tfl c3 c1; do timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C', d['ms_per_step'], d['value'])"
done | tee -a gpurun_out/r06_c29_c3w_wide.txt

#!/bin/bash
# round 6, call 14: scheduling switches under the recorded step (side-lane placement): FOCR_PARK_TAIL, FOCR_DEFER_SIDE, FOCR_MASK_EARLY, side stream off
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { timeout 600 env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$*', d['ms_per_step'], d['value'], d['config'].get('mode1_ms_per_step'))"; }
for rep in 1 2; do
run X=0
run FOCR_PARK_TAIL=1
run FOCR_PARK_TAIL=2
run FOCR_PARK_TAIL=3
run FOCR_DEFER_SIDE=1
run FOCR_WGRAD_SIDE=0
done | tee gpurun_out/r06_c14_sched.txt

#!/bin/bash
# round 6, call 62: the eager step of the final tree (what every rank of an N > 1 run launches): c3 at B = 128 and 16, FOCR_REPLAY=0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 128 16; do for R in 0 1; do
FOCR_REPLAY=$R timeout 600 python bench.py --config c3 --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 B=$B FOCR_REPLAY=$R', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r06_eager_vs_replay_final.txt

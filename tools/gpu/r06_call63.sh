#!/bin/bash
# round 6, call 63: the reference README's batch (16) on the final tree: c3, c1, tfl, sfl
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c3 c1 tfl sfl; do
timeout 600 python bench.py --config $C --batch 16 --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C B=16', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r06_batch16_final.txt

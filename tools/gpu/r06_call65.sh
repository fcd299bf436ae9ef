#!/bin/bash
# round 6, call 65: text-focus step with the HR branch of the criterion on the side stream, started before the SR network's forward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 128 16; do for M in 0 1 0 1; do
  FOCR_HR_SIDE=$M timeout 600 python bench.py --config tfl --batch $B --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs 2>gpurun_out/hr_side.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tfl B=$B FOCR_HR_SIDE=$M', d['ms_per_step'], d['final_loss'], d['config']['recorded_step'] and d['config']['recorded_step']['waits'])" || tail -3 gpurun_out/hr_side.err
done; done | tee gpurun_out/r06_hr_side_ab.txt

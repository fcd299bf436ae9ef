#!/bin/bash
# round 6, call 66: HR branch on the side stream as the default: focus tests, sfl / tfl numbers
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_text_focus.py tests/test_gpu_replay.py -m gpu -x -q 2>&1 | tail -3
for C in tfl sfl; do for B in 128 16; do for M in 0 1; do
  FOCR_HR_SIDE=$M timeout 600 python bench.py --config $C --batch $B --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C B=$B FOCR_HR_SIDE=$M', d['ms_per_step'])"
done; done; done | tee -a gpurun_out/r06_hr_side_ab.txt

#!/bin/bash
# round 6, call 48: attention backward variant by grid size: attention tests, small-batch steps
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_replay.py -m gpu -x -q -k "attention or recorded_step_equals" 2>&1 | tail -3
for C in c3 tfl; do for B in 8 16 24 32; do
  timeout 600 python bench.py --config $C --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C B=$B', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/r06_small_batch_attn_variant.txt

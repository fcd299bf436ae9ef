#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "train_mse_golden or traj3 or elementwise" 2>&1 | tail -2
for v in 1 0 1 0; do
  FOCR_DGRAD_FIRST=$v timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/c17_b_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/c17_b_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print("dgrad_first=$v", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done

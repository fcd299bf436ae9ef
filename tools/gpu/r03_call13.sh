#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 build/attn_ubench 128 2>&1 | grep -A1 PLANES | head -2 | tee gpurun_out/c13_attn.log
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "feature_enhancer or train_mse_golden or elementwise" 2>&1 | tail -2
for v in 1 0 1 0; do
  FOCR_ATTN_PLANES=$v timeout 200 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/c13_b_$v.log 2>&1
  python - <<PY
import json
for l in open('gpurun_out/c13_b_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print("planes=$v", d["value"], d["ms_per_step"], [(a["kernel"][:12], a.get("avg_launch_ms")) for a in d["roofline"]["also"] if "att" in a["kernel"]])
PY
done

#!/bin/bash
# parked weight gradients (issued beside the next attention backward): parity + same-box A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/c9
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "feature_enhancer or golden or oracle" > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log; tail -3 ${O}_pytest.log
for v in 1 0 1 0; do
  FOCR_DEFER_SIDE=$v timeout 200 python bench.py --no-cpu-baseline --steps 60 > ${O}_b_$v.log 2>&1
  python - <<PY
import json
for l in open('${O}_b_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('defer=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done

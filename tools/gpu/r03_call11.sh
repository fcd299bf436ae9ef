#!/bin/bash
# attention on pre-split planes: parity tests + same-box A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/c11
( timeout 200 build/fe_ubench 128 | tail -12 ) > ${O}_ubench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -x -q -m gpu -k "feature_enhancer or golden or oracle or attention" > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log; tail -4 ${O}_pytest.log
for v in 1 0 1 0; do
  FOCR_ATTN_PLANES=$v timeout 200 python bench.py --no-cpu-baseline --steps 60 > ${O}_b_$v.log 2>&1
  python - <<PY
import json
for l in open('${O}_b_$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('planes=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
done
cat gpurun_out/test_margins.txt 2>/dev/null | cut -c1-200 | tail -6

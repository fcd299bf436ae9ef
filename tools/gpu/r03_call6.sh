#!/bin/bash
# round 3, call 6: whole-SRB autograd node: model parity tests, host profile, same-box A/B (c3, c2)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/c6
rm -f gpurun_out/test_margins.txt
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -x -q -m gpu > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log
timeout 300 python tools/dev/host_profile.py > ${O}_host.log 2>&1
for v in 1 0 1; do
  FOCR_SRB_FUSED=$v timeout 200 python bench.py --no-cpu-baseline --steps 40 > ${O}_b_c3_srb$v.log 2>&1
  FOCR_SRB_FUSED=$v timeout 200 python bench.py --no-cpu-baseline --steps 40 --config c2 > ${O}_b_c2_srb$v.log 2>&1
  for c in c3 c2; do python - <<PY
import json
for l in open('${O}_b_${c}_srb$v.log'):
    if l.startswith('{'):
        d=json.loads(l); print('$c srb=$v', d['value'], d['ms_per_step'])
PY
  done
done
tail -5 ${O}_pytest.log; head -8 ${O}_host.log | tail -3

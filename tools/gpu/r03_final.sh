#!/bin/bash
# final check of HEAD: what the driver runs at round end (pytest -m gpu, smoke, default bench)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/final_pytest.log; tail -5 gpurun_out/final_pytest.log | head -3
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 20 > gpurun_out/final_bench.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/final_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config'].get('mode1_ms_per_step'), d.get('cpu_baseline',{}).get('value'))
PY

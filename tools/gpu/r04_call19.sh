#!/bin/bash
# full suite + smoke + all-configs bench + profile set r04c on the current tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04_c19_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r04_c19_pytest.log; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r04_c19_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --all-configs > gpurun_out/r04_c19_bench.log 2>gpurun_out/r04_c19_bench.err
python - <<PY
import json
for l in open('gpurun_out/r04_c19_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config'].get('name'), d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), d['config'].get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))
PY
bash tools/profile_round.sh r04c > gpurun_out/r04c_profile.log 2>&1; tail -3 gpurun_out/r04c_kt_total.txt; head -3 gpurun_out/r04c_gaps.txt

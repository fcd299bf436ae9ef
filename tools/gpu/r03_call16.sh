#!/bin/bash
# round 3: full GPU suite on the final code + final profile set (r03b) + bench lines of every configuration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/c16_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/c16_pytest.log; tail -6 gpurun_out/c16_pytest.log
bash tools/profile_round.sh r03b > gpurun_out/c16_profile.log 2>&1
timeout 900 python bench.py --all-configs > gpurun_out/c16_bench_all.log 2>&1
python - <<PY
import json
for l in open('gpurun_out/c16_bench_all.log'):
    if l.startswith('{'):
        d=json.loads(l); c=d.get('config',{}); print(c.get('name'), d.get('value'), d.get('ms_per_step'), d.get('roofline',{}).get('frac'), c.get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))
PY
python __graft_entry__.py > /dev/null 2>&1; python -c "import __graft_entry__ as G; G.smoke()" 2>&1 | tail -2

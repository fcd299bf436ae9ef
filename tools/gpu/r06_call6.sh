#!/bin/bash
# round 6, call 6: recorded step as the engine default: full `-m gpu` suite, smoke, host enqueue time, default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_c6_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r06_c6_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r06_c6_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 200 python tools/dev/host_time.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r06_c6_host_time.txt
timeout 900 python bench.py > gpurun_out/r06_c6_bench.log 2>gpurun_out/r06_c6_bench.err; tail -3 gpurun_out/r06_c6_bench.err
python - <<PY
import json
for l in open('gpurun_out/r06_c6_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], r.get('traffic'), d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'), d['config'].get('recorded_step'))
PY

#!/bin/bash
# round 6, call 15: the tree as committed: full `-m gpu` suite, `build(); smoke()`, default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06_c15_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r06_c15_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r06_c15_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
SECONDS=0
timeout 900 python bench.py > gpurun_out/r06_c15_bench.log 2>gpurun_out/r06_c15_bench.err; echo "bench wall ${SECONDS}s"; tail -3 gpurun_out/r06_c15_bench.err
python - <<PY
import json
for l in open('gpurun_out/r06_c15_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], r.get('traffic'), d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), d['config'].get('mode1_images_per_sec'), (d.get('cpu_baseline') or {}).get('value'), d['config'].get('recorded_step'))
PY

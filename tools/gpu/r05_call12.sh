#!/bin/bash
# round 5, call 12: launch + prologue cost of the chains; full GPU suite; smoke; default bench (CPU baseline legs, other
# configurations); profile set r05b (kernel trace, FETCH / WRITE / MFMA PMC passes, gaps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 200 build/fe_ubench 128 2>&1 | grep -E "prologue only" | tee gpurun_out/r05_c12_fe_prologue.txt
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r05_c12_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r05_c12_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r05_c12_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r05_c12_bench.log 2>gpurun_out/r05_c12_bench.err
python - <<PY
import json
for l in open('gpurun_out/r05_c12_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), d.get('cpu_baseline'))
        for a in r['also'][:8]: print('   ', a['kernel'][:50], a.get('frac'), a.get('avg_launch_ms', a.get('avg_call_ms')))
PY
bash tools/profile_round.sh r05b --no-other-configs > gpurun_out/r05b_profile.log 2>&1; tail -1 gpurun_out/r05b_kt_total.txt; head -1 gpurun_out/r05b_gaps.txt
head -8 gpurun_out/r05b_mfma_util.txt

#!/bin/bash
# frozen recognizer: eval-BatchNorm folded into the convolution in front of it (FOCR_CRNN_FOLD_BN): full suite, smoke, A/B,
# default bench, profile set r04h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04_c43_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r04_c43_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r04_c43_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in "fold FOCR_CRNN_FOLD_BN=1" "nofold FOCR_CRNN_FOLD_BN=0"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done
timeout 600 python bench.py > gpurun_out/r04_c43_bench.log 2>gpurun_out/r04_c43_bench.err
python - <<PY
import json
for l in open('gpurun_out/r04_c43_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))
PY
bash tools/profile_round.sh r04h --no-other-configs > gpurun_out/r04h_profile.log 2>&1; tail -1 gpurun_out/r04h_kt_total.txt; head -1 gpurun_out/r04h_gaps.txt

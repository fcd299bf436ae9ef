#!/bin/bash
# round 5, call 28: host micro-optimisations (plain-int pointer arguments, cached bound functions): full suite, host enqueue
# time, c2 / c3 timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r05_c28_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r05_c28_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r05_c28_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 200 python tools/dev/host_time.py 2>&1 | grep "host enqueue"
for r in 1 2 3; do timeout 300 python bench.py --config c2 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2', d['ms_per_step'], d['value'])"; done
for r in 1 2; do timeout 300 python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c3', d['ms_per_step'], d['value'])"; done

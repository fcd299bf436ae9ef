#!/bin/bash
# round 5, call 1: full GPU suite incl. the new full-size oracle tests (B = 128 / 64 / 32) and the eval-BatchNorm cache test,
# smoke, default bench (new roofline flop count, two-thread-count CPU baseline), c2 kernel trace (busy vs span)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu --durations=12 > gpurun_out/r05_c1_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r05_c1_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r05_c1_pytest.log | head -20
grep -A14 "slowest" gpurun_out/r05_c1_pytest.log | head -16
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r05_c1_bench.log 2>gpurun_out/r05_c1_bench.err
python - <<PY
import json
for l in open('gpurun_out/r05_c1_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), d.get('cpu_baseline'))
PY
# c2: un-profiled step time, then the kernel trace of the same command (busy = union of kernel intervals of all queues)
timeout 300 python bench.py --config c2 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2 un-profiled', d['ms_per_step'], d['value'])" | tee gpurun_out/r05_c2_unprofiled.txt
rocprofv3 --kernel-trace -d gpurun_out/p_r05_c2 -o bench -- python bench.py --config c2 --steps 12 --warmup 6 --no-cpu-baseline > gpurun_out/p_r05_c2.log 2>&1
DB=$(find gpurun_out/p_r05_c2 -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB clip_adam 8 > gpurun_out/r05_c2_gaps.txt; head -12 gpurun_out/r05_c2_gaps.txt
python tools/rocpd_bygrid.py $DB "" 8 > gpurun_out/r05_c2_bygrid.txt
python tools/rocpd_stats.py $DB gpurun_out/r05_c2_kernel_stats.csv 2> gpurun_out/r05_c2_kt_total.txt; cat gpurun_out/r05_c2_kt_total.txt
rm -rf gpurun_out/p_r05_c2
cat gpurun_out/test_margins.txt | grep -E "full_size|B = 32|sld_full" 

#!/bin/bash
# after the VALU work (pairwise splits, -fno-slp-vectorize, dQ pass select): kernel parity suite, bench, kernel-trace stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -4 > gpurun_out/c26_tests.log; tail -2 gpurun_out/c26_tests.log
for i in 1 2; do timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'])"; done | tee gpurun_out/c26_bench.log
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace -d gpurun_out/p_c26_kt -o bench -- $B > gpurun_out/p_c26_kt.log 2>&1
DB=$(find gpurun_out/p_c26_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r03c_bench_kernel_stats.csv 2> gpurun_out/r03c_kt_total.txt
python tools/rocpd_bygrid.py $DB "" 13 > gpurun_out/r03c_all_bygrid.txt; python tools/rocpd_gaps.py $DB clip_adam 6 > gpurun_out/r03c_gaps.txt
rm -rf gpurun_out/p_c26_kt; head -3 gpurun_out/r03c_gaps.txt

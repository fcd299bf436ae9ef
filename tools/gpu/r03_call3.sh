#!/bin/bash
# round 3, call 3: full GPU suite with the new tests (stroke-focus loss, gradient pins, full-size properties, dp_selfcheck
# identity + loader), bench (default line) and a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/c3
rm -f gpurun_out/test_margins.txt
timeout 1200 python -m pytest tests -q -m gpu > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log
timeout 300 python bench.py --no-cpu-baseline > ${O}_b_c3.log 2>&1
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace -d gpurun_out/p_c3_kt -o bench -- $B > ${O}_kt.log 2>&1
DB=$(find gpurun_out/p_c3_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB ${O}_kernel_stats.csv 2> ${O}_kt_total.txt
python tools/rocpd_bygrid.py $DB "" 13 > ${O}_all_bygrid.txt 2>&1
python tools/rocpd_gaps.py $DB clip_adam 6 > ${O}_gaps.txt 2>&1
rm -rf gpurun_out/p_c3_kt
tail -15 ${O}_pytest.log; cat gpurun_out/test_margins.txt; tail -c 400 ${O}_b_c3.log; head -12 ${O}_gaps.txt

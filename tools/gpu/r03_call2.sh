#!/bin/bash
# round 3, call 2: full GPU suite after the hardening changes + a kernel trace of the c3 step with the fused chains
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/c2
timeout 900 python -m pytest tests -x -q -m gpu > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace -d gpurun_out/p_c2_kt -o bench -- $B > ${O}_kt.log 2>&1
DB=$(find gpurun_out/p_c2_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB ${O}_kernel_stats.csv 2> ${O}_kt_total.txt
python tools/rocpd_bygrid.py $DB "" 13 > ${O}_all_bygrid.txt 2>&1
python tools/rocpd_gaps.py $DB > ${O}_gaps.txt 2>&1
rm -rf gpurun_out/p_c2_kt
tail -4 ${O}_pytest.log; head -40 ${O}_kernel_stats.csv | cut -c1-150

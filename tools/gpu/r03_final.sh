#!/bin/bash
# final check of HEAD: what the driver runs at round end (pytest -m gpu, build + smoke in one interpreter, default bench),
# plus the other configurations' lines and a kernel-trace summary of the final kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/final_pytest.log; grep -E "passed|failed|^FAILED|^E  " gpurun_out/final_pytest.log | head
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --all-configs > gpurun_out/final_bench.log 2>gpurun_out/final_bench.err
python - <<PY
import json
for l in open('gpurun_out/final_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(d['config'].get('workload', d['config'])[:40] if isinstance(d['config'].get('workload'), str) else d['config'], d.get('value'), d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), d['config'].get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))
PY
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace -d gpurun_out/p_fin_kt -o bench -- $B > gpurun_out/p_fin_kt.log 2>&1
DB=$(find gpurun_out/p_fin_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r03c_bench_kernel_stats.csv 2> gpurun_out/r03c_kt_total.txt
python tools/rocpd_bygrid.py $DB "" 13 > gpurun_out/r03c_all_bygrid.txt; python tools/rocpd_bygrid.py $DB conv3x3_halo 13 > gpurun_out/r03c_halo_bygrid.txt; python tools/rocpd_gaps.py $DB clip_adam 6 > gpurun_out/r03c_gaps.txt
rm -rf gpurun_out/p_fin_kt; head -2 gpurun_out/r03c_gaps.txt

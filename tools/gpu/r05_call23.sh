#!/bin/bash
# round 5, call 23: evidence lines of the final tree: the N > 1 bench line's new fields (two gloo ranks on the one device, the
# only multi-rank form a 1-GPU box can run), c2 un-profiled + busy under the profiler
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FOCR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 10 --batch 32 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r05_bench_line_2ranks_gloo.json
python -c "import json; d=json.load(open('gpurun_out/r05_bench_line_2ranks_gloo.json')); print({k: d.get(k) for k in ('n_gpus','value','ms_per_step','rccl_ranks','rccl_version','exposed_comm_ms','comm_path')})"
for r in 1 2 3; do timeout 300 python bench.py --config c2 --steps 60 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2 un-profiled', d['ms_per_step'], d['value'])"; done | tee gpurun_out/r05_c2_final_unprofiled.txt
rocprofv3 --kernel-trace -d gpurun_out/p_r05_c2 -o bench -- python bench.py --config c2 --steps 12 --warmup 6 --no-cpu-baseline > gpurun_out/p_r05_c2.log 2>&1
DB=$(find gpurun_out/p_r05_c2 -name "*.db" | head -1)
python tools/rocpd_gaps.py $DB clip_adam 8 > gpurun_out/r05_c2_final_gaps.txt; head -1 gpurun_out/r05_c2_final_gaps.txt
rm -rf gpurun_out/p_r05_c2

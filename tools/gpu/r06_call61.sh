#!/bin/bash
# round 6, call 61: the N > 1 path of the final tree on the only multi-rank form a 1-GPU box can run (two gloo ranks on one device):
# functional check (eager steps, bucketed all-reduce), c3 and tfl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c3 tfl; do
FOCR_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --config $C --steps 10 --warmup 5 --batch 16 2>gpurun_out/r06_2ranks_$C.err | grep "^{" | tail -1 > gpurun_out/r06_bench_line_2ranks_gloo_$C.json
python -c "
import json; d=json.load(open('gpurun_out/r06_bench_line_2ranks_gloo_$C.json')); print('$C', {k: d.get(k) for k in ('n_gpus','value','ms_per_step','rccl_ranks','exposed_comm_ms','comm_path','final_loss')}, d['config'].get('recorded_step'))" || tail -5 gpurun_out/r06_2ranks_$C.err
done

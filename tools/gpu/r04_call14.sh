#!/bin/bash
# weight gradients on the side stream vs in line (FOCR_WGRAD_SIDE=0), interleaved, with the single-pass attention backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c14.log
run() { env $2 timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c14.log; }
for rep in 1 2; do
  run side-on FOCR_WGRAD_SIDE=1
  run side-off FOCR_WGRAD_SIDE=0
done

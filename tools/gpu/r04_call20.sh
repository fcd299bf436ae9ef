#!/bin/bash
# ordering switches re-measured with the single-pass attention backward (which owns a whole CU: 138 KB of LDS, 8 waves)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c20.log
run() { env $2 timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs $3 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c20.log; }
for rep in 1 2; do
  run default X=1
  run fe-wgrad-late FOCR_FE_WGRAD_EARLY=0
  run defer-side FOCR_DEFER_SIDE=1
  run dgrad-second FOCR_DGRAD_FIRST=0
  run fwd-variant-2 X=1 "--tuning 1=2"
  run fwd-variant-0 X=1 "--tuning 1=0"
done

#!/bin/bash
# round 6, call 59: block timeline of the halo kernel after its spills were removed (64 -> 64, B = 128): with / without residual
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for NR in "" 1; do
  if [ -n "$NR" ]; then export H3_UB_NORES=1; else unset H3_UB_NORES; fi
  H3_TRACE_RUN=3 timeout 120 build/conv_ubench_trace 128 "srb 3x3" > gpurun_out/c59_trace_$NR.log 2>&1
  echo "=== residual: $([ -n "$NR" ] && echo no || echo yes)"; python tools/dev/halo_trace_summary.py gpurun_out/c59_trace_$NR.log 2>&1 | grep -v "histogram" | head -24
done | tee gpurun_out/r06_halo_timeline.txt

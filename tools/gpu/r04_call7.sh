#!/bin/bash
# single-pass attention backward v3 (hand-ordered software pipeline) + skewed staging
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c7.log
for v in "" PRIO DQ STAGE; do
  echo "== variant: ${v:-default}" | tee -a gpurun_out/r04_c7.log
  timeout 120 build/attn_ubench_b1$v 128 1 2 b1 2>&1 | grep -E "^bwd1" | tee -a gpurun_out/r04_c7.log
done
timeout 120 build/attn_ubench_b1 8 1 2 b1 2>&1 | grep -E "^bwd1" | tee -a gpurun_out/r04_c7.log

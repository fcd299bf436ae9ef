#!/bin/bash
# single-pass attention backward v2 (dQ product merged into the next tile's code, loads a whole iteration ahead)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c4_abl.log
for v in "" STAGE DQ; do
  echo "== ablation: ${v:-none}" | tee -a gpurun_out/r04_c4_abl.log
  timeout 120 build/attn_ubench_b1$v 128 1 2 b1 2>&1 | grep -E "^bwd1" | tee -a gpurun_out/r04_c4_abl.log
done
timeout 120 build/attn_ubench_b1 8 1 2 b1 2>&1 | grep -E "^bwd1" | tee -a gpurun_out/r04_c4_abl.log

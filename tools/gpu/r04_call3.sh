#!/bin/bash
# single-pass attention backward: phase ablations (timing only; ablated builds compute wrong results by construction)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c3_abl.log
for v in "" STAGE MFMA12 VALU T BAR DQ; do
  echo "== ablation: ${v:-none}" | tee -a gpurun_out/r04_c3_abl.log
  timeout 120 build/attn_ubench_b1$v 128 1 2 b1 2>&1 | grep -E "^bwd1" | tee -a gpurun_out/r04_c3_abl.log
done

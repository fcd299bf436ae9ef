#!/bin/bash
# alternate code paths behind the A/B switches stay green
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for env in "FOCR_ATTN_PLANES=1" "FOCR_SRB_FUSED=0" "FOCR_FE_FUSED=0" "FOCR_DEFER_SIDE=1 FOCR_FE_WGRAD_EARLY=0 FOCR_DGRAD_FIRST=0"; do
  echo "== $env"
  env $env timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -2
done 2>&1 | tee gpurun_out/c18_switches.log

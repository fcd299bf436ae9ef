#!/bin/bash
# round 5, call 9: fixed (per-launch) cost of the FeatureEnhancer chains: kernel time against tiles per block (B = 8 = one 32-row
# tile per wave-slot ... B = 128 = 16), i.e. the weight-staging prologue
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 2 8 16 32 64 128; do
  echo "== B=$b"; timeout 120 build/fe_ubench $b 2>&1 | grep -E "^fe_" 
done | tee gpurun_out/r05_c9_fe_prologue.txt

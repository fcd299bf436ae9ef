#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 0 4 8 0 4; do echo "== touch $d"; timeout 300 build/attn_ubench_t$d 128 2>&1 | grep -E "^fwd|PACKED"; done | tee gpurun_out/c15_touch.log

#!/bin/bash
# dQ pass: -D folded into the accumulator + select (sel) vs the previous form (noslp), standalone, interleaved
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do for v in noslp sel; do echo "== $v"; timeout 300 build/attn_ubench_$v 128 2>&1 | grep -E "^bwd" | sed 's/max diff.*//'; done; done | tee gpurun_out/c27_dq.log

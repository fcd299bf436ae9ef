#!/bin/bash
# round 5, call 13: wave stagger in the FeatureEnhancer chains (upper four waves delayed by n us after the prologue)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for n in 0 3 6 9 12 0; do echo "== stagger $n us"; timeout 200 build/fe_ubench_s$n 128 2>&1 | grep -E "^fe_|FAIL"; done | tee gpurun_out/r05_c13_stagger.txt

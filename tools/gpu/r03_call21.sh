#!/bin/bash
# halo staging as ONE batch of 13 loads per thread instead of 7 + 6 (is the staging phase latency- or bandwidth-bound?).
# conv_ubench_1b / conv_ubench_trace1b were built from a tree with that variant behind -DH3_ONE_BATCH (since removed: slower)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
echo "== two batches"; timeout 60 build/conv_ubench 128 "srb 3x3" | grep srb | sed 's/.*halo x3/halo x3/'
echo "== one batch";   timeout 60 build/conv_ubench_1b 128 "srb 3x3" | grep srb | sed 's/.*halo x3/halo x3/'
} > gpurun_out/c21_onebatch.log 2>&1
H3_TRACE_RUN=1 timeout 60 build/conv_ubench_trace1b 128 "srb 3x3" > gpurun_out/c21_trace_onebatch.log 2>&1
cat gpurun_out/c21_onebatch.log

#!/bin/bash
# halo kernel block timeline (wall-clock stamps per block and phase) for one isolated launch and for a launch in a chain
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
H3_TRACE_RUN=1 timeout 60 build/conv_ubench_trace 128 "srb 3x3" > gpurun_out/c20_trace_single.log 2>&1
H3_TRACE_RUN=3 timeout 60 build/conv_ubench_trace 128 "srb 3x3" > gpurun_out/c20_trace_chain.log 2>&1
grep TRACE gpurun_out/c20_trace_*.log

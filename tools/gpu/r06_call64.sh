#!/bin/bash
# round 6, call 64: soak of the recorded text-focus step with labels that change every step (several capacity buckets)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=16 STEPS=400 timeout 900 python tools/dev/tfl_soak.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_tfl_soak.txt

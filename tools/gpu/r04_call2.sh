#!/bin/bash
# single-pass attention backward: first run (correctness vs the two-pass kernels + timing), B = 8 then B = 128
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 build/attn_ubench 8 2>&1 | grep -E "^bwd" | tee gpurun_out/r04_c2_attn.log
timeout 300 build/attn_ubench 128 2>&1 | grep -E "^bwd" | tee -a gpurun_out/r04_c2_attn.log

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 build/attn_ubench 128 2>&1 | grep -E "^bwd|PACKED|bwd \(no" | tee gpurun_out/c15_pitch.log

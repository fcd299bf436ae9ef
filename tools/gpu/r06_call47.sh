#!/bin/bash
# round 6, call 47: attention backward, single pass vs two passes against the batch
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/dev/attn_bwd_batch.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_attn_bwd_batch.txt

#!/bin/bash
# ds_read_b64_tr_b16 lane semantics probe + same-box baseline of the two-pass attention backward
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 60 build/tr_probe > gpurun_out/r04_tr_probe.txt 2>&1
timeout 300 build/attn_ubench 128 2>&1 | grep -E "^fwd|^bwd" | tee gpurun_out/r04_c1_attn.log

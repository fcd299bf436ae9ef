#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 build/attn_ubench 128 > gpurun_out/c10_attn.log 2>&1; echo "rc=$?" >> gpurun_out/c10_attn.log
cat gpurun_out/c10_attn.log

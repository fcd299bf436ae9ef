#!/bin/bash
# attention kernels after the pairwise split / packed-fp32 VALU rewrite: timing (standalone) + parity tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 build/attn_ubench 128 2>&1 | grep -E "^fwd|^bwd" | tee gpurun_out/c22_attn.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention or fe_chain or feature_enhancer" 2>&1 | tail -3 | tee -a gpurun_out/c22_attn.log

#!/bin/bash
# round 6, call 53: gemm_big: next chunk's split + LDS store between the two k-steps (under the first k-step's MFMAs) vs after both
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_gbmid.so "" fudanocr_amd/libfocr_hip_gbmid.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/gemm_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_gemm_big_store_mid.txt

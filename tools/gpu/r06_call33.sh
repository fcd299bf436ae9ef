#!/bin/bash
# round 6, call 33: halo kernel, split products: three dependent MFMAs per accumulator in a row (default) vs alternating accumulators
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_h3il.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/halo_bench.py; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_halo_interleave_ab.txt

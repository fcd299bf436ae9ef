#!/bin/bash
# round 6, call 40: halo kernel on multi-slice layers: second resident block of a CU delayed by H3_STAGGER x 0.43 us (does breaking
# the lock step of the two co-resident blocks overlap one's staging with the other's contraction?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_stg6.so fudanocr_amd/libfocr_hip_stg12.so fudanocr_amd/libfocr_hip_stg24.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | head -4; done | tee gpurun_out/r06_halo_stagger_multislice.txt

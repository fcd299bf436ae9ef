#!/bin/bash
# round 6, call 60: stagger of the second resident block again, now that the kernel is spill-free and its epilogue is 2 us (block life:
# stage 5.6 + contract 6.9 + epilogue 2.1 us, both resident blocks in the same phase 41 % of the time)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_stg4.so fudanocr_amd/libfocr_hip_stg8.so fudanocr_amd/libfocr_hip_stg14.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | tail -9; done | tee gpurun_out/r06_halo_stagger_again.txt

#!/bin/bash
# round 6, call 44: small grids (one block per CU): weight-chunk prefetch distance 2 (default) / 3 / 4, loader wave off
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_dist3.so fudanocr_amd/libfocr_hip_dist4.so; do echo "== FOCR_LIB=$L (FOCR_H3_LOADER=0)"; FOCR_H3_LOADER=0 FOCR_LIB=$L timeout 300 python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | head -7; done | tee gpurun_out/r06_halo_small_grid_dist.txt

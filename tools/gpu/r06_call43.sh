#!/bin/bash
# round 6, call 43: loader-wave form of the halo kernel: correctness (halo + conv tests), then A/B: never / small grids
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or conv2d" 2>&1 | tail -4
for M in 0 256; do echo "== FOCR_H3_LOADER=$M"; FOCR_H3_LOADER=$M timeout 300 python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | head -5; done | tee gpurun_out/r06_halo_loader_ab.txt

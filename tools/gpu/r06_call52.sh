#!/bin/bash
# round 6, call 52: large plain GEMMs, generic kernel vs 256 x 128 tiles
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for M in 0 1; do echo "== FOCR_GEMM_BIG=$M"; FOCR_GEMM_BIG=$M python tools/dev/gemm_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_gemm_big_standalone.txt

#!/bin/bash
# halo kernel: weight-chunk prefetch distance experiment (H3_DIST = 2 / 4 / 6), SRB shapes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 2 4 6; do echo "== H3_DIST=$d"; timeout 120 build/conv_ubench_d$d 128 "64->64"; done > gpurun_out/c7_halo_dist.log 2>&1
cat gpurun_out/c7_halo_dist.log

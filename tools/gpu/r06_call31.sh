#!/bin/bash
# round 6, call 31: halo kernel block order on multi-slice layers: output group fastest inside an XCD (FOCR_H3_GROUP_FAST) A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for M in 0 1; do echo "== FOCR_H3_GROUP_FAST=$M"; FOCR_H3_GROUP_FAST=$M python tools/dev/halo_bench.py; done 2>&1 | tee gpurun_out/r06_halo_group_fast_ab.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for a in d2 abl_MFMA abl_STAGE abl_EPI abl_SE abl_ALL; do echo "== $a"; timeout 120 build/conv_ubench_$a 128 "srb 3x3" | grep srb | sed 's/.*halo x3/halo x3/'; done > gpurun_out/c8_halo_abl.log 2>&1
cat gpurun_out/c8_halo_abl.log

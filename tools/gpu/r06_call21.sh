#!/bin/bash
# round 6, call 21: conv_wgrad_bx3_wide_kernel standalone at the SLD shapes + ablations (1 = no global loads after the first chunk, 2 = no MFMAs)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "" wgw1 wgw2; do
  L=fudanocr_amd/libfocr_hip${v:+_$v}.so
  echo "== $L"; FOCR_LIB=$PWD/$L timeout 200 python tools/dev/wgrad_wide_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_c21_wgw.txt

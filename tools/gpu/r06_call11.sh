#!/bin/bash
# round 6, call 11: where the GRU backward's 8 us per step go: ablations (1 = no stores, 2 = no loads after step 1, 4 = no MFMA chain)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "" gruabl1 gruabl2 gruabl4 gruabl7; do
  L=fudanocr_amd/libfocr_hip${v:+_$v}.so
  echo "== $L"; FOCR_LIB=$PWD/$L timeout 200 python tools/dev/gru_bench.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r06_c11_gru_abl.txt

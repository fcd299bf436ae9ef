#!/bin/bash
# round 6, call 57: single-product halo launches without a residual: three blocks per CU (161 VGPRs, 50.7 KB of LDS) vs two
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_occ3.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/ | planes 2:.* | planes 1:/ | planes 1:/'; done | tee gpurun_out/r06_halo_occ3_ab.txt
for C in c3 tfl c5; do for L in "" fudanocr_amd/libfocr_hip_occ3.so "" fudanocr_amd/libfocr_hip_occ3.so; do
  FOCR_LIB=$L timeout 600 python bench.py --config $C --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', '$L' or 'default (two blocks per CU)', d['ms_per_step'])"
done; done | tee -a gpurun_out/r06_halo_occ3_ab.txt

#!/bin/bash
# round 6, call 58: halo staging as ONE batch of 13 loads per thread (fits now that the kernel no longer spills) vs 7 + 6
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in "" fudanocr_amd/libfocr_hip_onebatch.so; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | tail -9; done | tee gpurun_out/r06_halo_one_batch_ab.txt
for C in c3 tfl c5; do for L in "" fudanocr_amd/libfocr_hip_onebatch.so "" fudanocr_amd/libfocr_hip_onebatch.so; do
  FOCR_LIB=$L timeout 600 python bench.py --config $C --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', '$L' or 'default (7 + 6)', d['ms_per_step'])"
done; done | tee -a gpurun_out/r06_halo_one_batch_ab.txt

#!/bin/bash
# round 6, call 56: halo kernel, residual operand requested AFTER the contraction (no registers across it: every instantiation spill-free)
# vs before the last slice (default)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FOCR_LIB=fudanocr_amd/libfocr_hip_lateres.so timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or conv2d" 2>&1 | tail -2
for C in c3 tfl sfl c5 c1; do for L in "" fudanocr_amd/libfocr_hip_lateres.so "" fudanocr_amd/libfocr_hip_lateres.so; do
  FOCR_LIB=$L timeout 600 python bench.py --config $C --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', '$L' or 'default (early residual)', d['ms_per_step'])"
done; done | tee gpurun_out/r06_halo_late_res_ab.txt

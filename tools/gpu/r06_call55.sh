#!/bin/bash
# round 6, call 55: halo kernel instantiation WITHOUT the residual registers (161-179 VGPRs, no scratch) for launches that have none
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or conv2d" 2>&1 | tail -2
for M in 0 1; do echo "== FOCR_H3_NORES=$M"; FOCR_H3_NORES=$M python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_halo_nores_ab.txt
for C in c3 tfl c5 c1; do for M in 0 1 0 1; do
  FOCR_H3_NORES=$M timeout 600 python bench.py --config $C --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C FOCR_H3_NORES=$M', d['ms_per_step'])"
done; done | tee -a gpurun_out/r06_halo_nores_ab.txt

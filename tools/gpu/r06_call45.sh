#!/bin/bash
# round 6, call 45: single-product halo launches: weight-chunk prefetch distance 5 (six 4 KB buffers) vs 2: tests, micro-benchmark, c5 / tfl
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo or conv2d" 2>&1 | tail -3
for L in fudanocr_amd/libfocr_hip_d1_2.so ""; do echo "== FOCR_LIB=$L"; FOCR_LIB=$L timeout 300 python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/ | planes 2:.* | planes 1:/ | planes 1:/'; done | tee gpurun_out/r06_halo_dist1_ab.txt
for C in c5 tfl c3; do for L in fudanocr_amd/libfocr_hip_d1_2.so "" fudanocr_amd/libfocr_hip_d1_2.so ""; do
  FOCR_LIB=$L timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', '$L' or 'default(dist 5)', d['ms_per_step'])"
done; done | tee -a gpurun_out/r06_halo_dist1_ab.txt

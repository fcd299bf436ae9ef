#!/bin/bash
# round 6, call 54: after making the mask epilogue a compile-time variant (the run-time form spilled in EVERY halo launch): halo tests,
# halo micro-benchmark, masked vs unmasked focus steps, c3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo" 2>&1 | tail -2
python tools/dev/halo_bench.py 2>&1 | grep -v amdgpu.ids | tail -9 | tee gpurun_out/r06_halo_after_mask_template.txt
for C in tfl sfl; do for M in 0 1 0 1; do
  FOCR_HALO_MASK=$M timeout 600 python bench.py --config $C --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C FOCR_HALO_MASK=$M', d['ms_per_step'])"
done; done | tee -a gpurun_out/r06_halo_after_mask_template.txt
for C in c3 c5; do timeout 600 python bench.py --config $C --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C', d['ms_per_step'])"; done | tee -a gpurun_out/r06_halo_after_mask_template.txt

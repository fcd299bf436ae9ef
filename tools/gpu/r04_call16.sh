#!/bin/bash
# 3x3 weight gradient: partial tiles folded in fixed order (no atomics) vs the atomic group reduce, parity + in-step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c16.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3 | tee -a gpurun_out/r04_c16.log
run() { env $2 timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c16.log; }
for rep in 1 2 3; do
  run fold FOCR_X=1
  run atomic FOCR_LIB=$PWD/fudanocr_amd/libfocr_hip_c3old.so
done

#!/bin/bash
# round 6, call 51: 256 x 128 tile GEMM for the recognizers' large 1 x 1 layers: conv tests, tfl / c5 A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "conv2d" 2>&1 | tail -3
for C in tfl c5 sfl; do for M in 0 1 0 1; do
  FOCR_GEMM_BIG=$M timeout 600 python bench.py --config $C --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C FOCR_GEMM_BIG=$M', d['ms_per_step'])"
done; done | tee gpurun_out/r06_gemm_big_ab.txt

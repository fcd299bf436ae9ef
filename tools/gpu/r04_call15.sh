#!/bin/bash
# bf16x3 weight gradient of the 9x9 output layer: parity, then in-step A/B against the fp32-MFMA kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c15.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv9x9" 2>&1 | tail -3 | tee -a gpurun_out/r04_c15.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or gradients" 2>&1 | tail -3 | tee -a gpurun_out/r04_c15.log
run() { env $2 timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c15.log; }
for rep in 1 2; do
  run c9-bx3 FOCR_C9_WGRAD_BX3=1
  run c9-fp32 FOCR_C9_WGRAD_BX3=0
done

#!/bin/bash
# relu backward folded into the maxpool backward (CRNN conv0/1/3/5): parity + step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c23.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "maxpool or pool or lstm" 2>&1 | tail -3 | tee -a gpurun_out/r04_c23.log
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -3 | tee -a gpurun_out/r04_c23.log
for rep in 1 2; do timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['value'], r['final_loss'])" | tee -a gpurun_out/r04_c23.log; done

#!/bin/bash
# single-pass attention backward as the library default: kernel + model parity, then the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention or feature_enhancer or fe_chain" 2>&1 | tail -5 | tee gpurun_out/r04_c8.log
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -m gpu 2>&1 | tail -5 | tee -a gpurun_out/r04_c8.log
timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('single-pass', r['ms_per_step'], r['value']); [print(a['kernel'][:60], a.get('avg_launch_ms')) for a in r['roofline']['also']]" | tee -a gpurun_out/r04_c8.log
timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs --tuning 3=1 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('two-pass', r['ms_per_step'], r['value']); [print(a['kernel'][:60], a.get('avg_launch_ms')) for a in r['roofline']['also']]" | tee -a gpurun_out/r04_c8.log
timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('single-pass', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c8.log
timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs --tuning 3=1 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('two-pass', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c8.log

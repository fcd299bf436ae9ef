#!/bin/bash
# default bench line with config.other_configs; new conv-relu-pool kernel test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv_relu_maxpool or maxpool" 2>&1 | tail -25
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 20 > gpurun_out/r04_c24_bench.log 2> gpurun_out/r04_c24_bench.err; echo "bench wall ${SECONDS}s"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/r04_c24_bench.log') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['config']['other_configs'], d['config']['mode1_ms_per_step'], d['cpu_baseline']['value'])
PY

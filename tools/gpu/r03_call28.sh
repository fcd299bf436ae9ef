#!/bin/bash
# final form of the VALU work: kernel + model parity, default bench (incl. the mode-1 leg)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/c28_tests.log; tail -2 gpurun_out/c28_tests.log
for i in 1 2; do timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('mode1_ms_per_step'))"; done | tee gpurun_out/c28_bench.log

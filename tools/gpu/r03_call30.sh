#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 > gpurun_out/c30_tests.log; grep -E "^FAILED|^E  |passed|failed" gpurun_out/c30_tests.log | head -20
timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('bench', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('mode1_ms_per_step'))"

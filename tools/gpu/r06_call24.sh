#!/bin/bash
# round 6, call 24: K = 64 weight gradients (GRU W_ih, 1x1 conv, W_hh cross product) on the streaming kernel: tests, c1 A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "gru or linear or conv" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_models.py -q -k "tsrn" 2>&1 | tail -4
for Q in 1 0 1 0; do FOCR_LW_QUART=$Q timeout 600 python bench.py --config c1 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c1 quart=$Q', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r06_c24_c1.txt

#!/bin/bash
# round 6, call 18: BatchNorm backward reduce with 16-row slabs on wide short tensors: tests, c5 / tfl / c3 timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "bn or batchnorm or norm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_sld.py -q -m gpu 2>&1 | tail -2
for C in c5 c5 tfl c3; do timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C', d['ms_per_step'], d['value'], d['roofline']['step_algorithmic_tflops'])"
done | tee gpurun_out/r06_c18.txt

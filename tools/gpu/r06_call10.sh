#!/bin/bash
# round 6, call 10: recorded step for the SLD engine (c5): tests, c5 timing replay vs eager
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py -q -k "sld" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_sld.py -q -m gpu 2>&1 | tail -4
for R in 1 0; do FOCR_REPLAY=$R timeout 600 python bench.py --config c5 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c5 replay=$R', d['ms_per_step'], d['value'], d['config'].get('recorded_step'))"
done | tee gpurun_out/r06_c10_c5.txt

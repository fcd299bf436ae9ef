#!/bin/bash
# split helper: volatile vs plain asm guard, interleaved on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do for v in vol plain; do
  lib=fudanocr_amd/libfocr_hip_vol.so; [ $v = plain ] && lib=fudanocr_amd/libfocr_hip.so
  FOCR_LIB=$PWD/$lib timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['ms_per_step'], d['value'], d['config'].get('mode1_ms_per_step'))"
done; done | tee gpurun_out/c32_ab.log

#!/bin/bash
# whole-step A/B of the pairwise split helpers and of -fno-slp-vectorize (no v_pk_* f32): base = previous commit's kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2 3; do for v in base new noslp; do
  lib=fudanocr_amd/libfocr_hip_$v.so; [ $v = new ] && lib=fudanocr_amd/libfocr_hip.so
  FOCR_LIB=$PWD/$lib timeout 300 python bench.py --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['ms_per_step'], d['value'])"
done; done | tee gpurun_out/c24_ab.log
FOCR_LIB=$PWD/fudanocr_amd/libfocr_hip_noslp.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/c24_ab.log

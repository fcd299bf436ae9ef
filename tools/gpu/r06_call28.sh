#!/bin/bash
# round 6, call 28: row-streaming 3x3 weight gradient generalised to wide layers on narrow maps (images abreast): conv tests, standalone A/B, c5 A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "test_conv2d" 2>&1 | tail -6
for X in 1 0; do echo "== FOCR_C3W_WIDE=$X"; FOCR_C3W_WIDE=$X timeout 300 python tools/dev/wgrad_wide_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_c28_c3w_wide.txt
for X in 1 0 1 0; do FOCR_C3W_WIDE=$X timeout 600 python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c5 c3w_wide=$X', d['ms_per_step'], d['value'])"
done | tee -a gpurun_out/r06_c28_c3w_wide.txt
timeout 900 python -m pytest tests/test_sld.py -q -m gpu 2>&1 | tail -2

#!/bin/bash
# round 6, call 22: XCD-aware pixel splits in the wide weight-gradient kernel: standalone A/B, c5 step A/B, SLD full-size test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for X in 0 1 2; do echo "== FOCR_WGW_XCD=$X"; FOCR_WGW_XCD=$X timeout 300 python tools/dev/wgrad_wide_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r06_c22_wgw_xcd.txt
for X in 0 1 2 0 1; do FOCR_WGW_XCD=$X timeout 600 python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c5 xcd=$X', d['ms_per_step'], d['value'])"
done | tee -a gpurun_out/r06_c22_wgw_xcd.txt
timeout 900 python -m pytest tests/test_sld.py -q -m gpu 2>&1 | tail -2

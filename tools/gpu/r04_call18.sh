#!/bin/bash
# 9x9 output-layer weight gradient with next-row prefetch and division-free staging: parity + serial kernel trace row
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv9x9" 2>&1 | tail -2
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs"
FOCR_WGRAD_SIDE=0 rocprofv3 --kernel-trace -d gpurun_out/p_serial_kt -o bench -- $B > gpurun_out/p_serial_kt.log 2>&1
DB=$(find gpurun_out/p_serial_kt -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "conv9x9" 13 | cut -c1-110
rm -rf gpurun_out/p_serial_kt
for rep in 1 2; do timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['value'])"; done

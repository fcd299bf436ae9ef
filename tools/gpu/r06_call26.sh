#!/bin/bash
# round 6, call 26: c5 and c3 on ONE queue (standalone kernel durations)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c5 c3; do
FOCR_WGRAD_SIDE=0 rocprofv3 --kernel-trace -d gpurun_out/p_$C -o t -- python bench.py --config $C --steps 8 --warmup 6 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_$C -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" 16 > gpurun_out/r06_${C}_serial_bygrid.txt; head -28 gpurun_out/r06_${C}_serial_bygrid.txt
rm -rf gpurun_out/p_$C
done

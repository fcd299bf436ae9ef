#!/bin/bash
# round 6, call 13: c1 kernel breakdown with the loader / compute GRU scans
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
C=c1
rocprofv3 --kernel-trace -d gpurun_out/p_$C -o t -- python bench.py --config $C --steps 8 --warmup 6 --no-cpu-baseline --no-other-configs > gpurun_out/r06_c13_${C}_prof.log 2>&1
DB=$(find gpurun_out/p_$C -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" 16 > gpurun_out/r06b_${C}_bygrid.txt; head -30 gpurun_out/r06b_${C}_bygrid.txt
python tools/rocpd_gaps.py $DB clip_adam 6 > gpurun_out/r06b_${C}_gaps.txt; head -4 gpurun_out/r06b_${C}_gaps.txt
rm -rf gpurun_out/p_$C

#!/bin/bash
# round 6, call 9: current-tree kernel breakdowns of c5 (SLD) and c1 (TSRN)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c5; do
rocprofv3 --kernel-trace -d gpurun_out/p_$C -o t -- python bench.py --config $C --steps 8 --warmup 6 --no-cpu-baseline > gpurun_out/r06_c9_${C}_prof.log 2>&1
DB=$(find gpurun_out/p_$C -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" 14 > gpurun_out/r06c_${C}_bygrid.txt; head -16 gpurun_out/r06c_${C}_bygrid.txt
M=clip_adam; [ $C = c5 ] && M=adadelta
python tools/rocpd_gaps.py $DB $M 6 > gpurun_out/r06c_${C}_gaps.txt; head -2 gpurun_out/r06c_${C}_gaps.txt
rm -rf gpurun_out/p_$C
done

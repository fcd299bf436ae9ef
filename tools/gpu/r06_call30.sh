#!/bin/bash
# round 6, call 30: final-tree kernel breakdowns of c1, c5, c2, tfl (two queues, as benched)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c1 c5 c2 tfl; do
S=8; W=6; [ $C = tfl ] && S=5 && W=4
rocprofv3 --kernel-trace -d gpurun_out/p_$C -o t -- python bench.py --config $C --steps $S --warmup $W --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_$C -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06d_${C}_bygrid.txt; head -12 gpurun_out/r06d_${C}_bygrid.txt
rm -rf gpurun_out/p_$C
done

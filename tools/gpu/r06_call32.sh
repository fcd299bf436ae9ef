#!/bin/bash
# round 6, call 32: c3 kernel table with everything on ONE queue (FOCR_WGRAD_SIDE=0): every kernel at its un-overlapped in-step cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=8; W=6
FOCR_WGRAD_SIDE=0 rocprofv3 --kernel-trace -d gpurun_out/p_one -o t -- python bench.py --config c3 --steps $S --warmup $W --no-cpu-baseline --no-other-configs > gpurun_out/p_one.log 2>&1
tail -1 gpurun_out/p_one.log | cut -c1-300
DB=$(find gpurun_out/p_one -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06d_c3_one_queue_bygrid.txt; head -45 gpurun_out/r06d_c3_one_queue_bygrid.txt; tail -2 gpurun_out/r06d_c3_one_queue_bygrid.txt
rm -rf gpurun_out/p_one

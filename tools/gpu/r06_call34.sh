#!/bin/bash
# round 6, call 34: the --mask variant of c3 (four channels: both 9x9 layers on the generic kernels): step time and kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --config c3 --mask --steps 40 --warmup 10 --no-other-configs > gpurun_out/c3_mask.log 2>&1; tail -1 gpurun_out/c3_mask.log | cut -c1-400
python bench.py --config c3 --steps 40 --warmup 10 --no-other-configs --no-cpu-baseline > gpurun_out/c3_nomask.log 2>&1; tail -1 gpurun_out/c3_nomask.log | cut -c1-200
S=8; W=6
rocprofv3 --kernel-trace -d gpurun_out/p_m -o t -- python bench.py --config c3 --mask --steps $S --warmup $W --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_m -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06d_c3_mask_bygrid.txt; head -30 gpurun_out/r06d_c3_mask_bygrid.txt
rm -rf gpurun_out/p_m

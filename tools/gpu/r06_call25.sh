#!/bin/bash
# round 6, call 25: tfl kernel breakdown at B = 128 after the BatchNorm fold
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/p_tfl -o t -- python bench.py --config tfl --steps 6 --warmup 4 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_tfl -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" 10 > gpurun_out/r06b_tfl_bygrid.txt; head -36 gpurun_out/r06b_tfl_bygrid.txt
python tools/rocpd_gaps.py $DB clip_adam 4 > gpurun_out/r06b_tfl_gaps.txt; head -2 gpurun_out/r06b_tfl_gaps.txt
rm -rf gpurun_out/p_tfl

#!/bin/bash
# one steady-state step in launch order (kernel trace): where the small copies / fills sit, which queue runs what
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/p_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_seq.log 2>&1
DB=$(find gpurun_out/p_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r04_step_sequence.txt
python tools/rocpd_gaps.py $DB clip_adam 6 | head -5
wc -l gpurun_out/r04_step_sequence.txt; rm -rf gpurun_out/p_seq

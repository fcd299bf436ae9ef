#!/bin/bash
# round 6, call 16: the replayed c3 step in launch order (both queues): where the low-occupancy time is
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N_LANES=2 PHASE=replay rocprofv3 --kernel-trace -d gpurun_out/p_seq -o t -- python tools/dev/replay_trace.py 128 > gpurun_out/r06_c16.log 2>&1
DB=$(find gpurun_out/p_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 3 > gpurun_out/r06_step_sequence.txt
wc -l gpurun_out/r06_step_sequence.txt; tail -1 gpurun_out/r06_c16.log
rm -rf gpurun_out/p_seq

#!/bin/bash
# round 6, call 2: why the replayed step is slower on the GPU than the eager one: kernel traces of both, single-lane replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in 1 2 6; do N_LANES=$L PHASE=replay timeout 300 python tools/dev/replay_trace.py 128 2>&1 | tail -1; done | tee gpurun_out/r06_c2_lanes.txt
PHASE=eager timeout 300 python tools/dev/replay_trace.py 128 2>&1 | tail -1 | tee -a gpurun_out/r06_c2_lanes.txt
for P in eager replay; do
  PHASE=$P rocprofv3 --kernel-trace -d gpurun_out/p_$P -o t -- python tools/dev/replay_trace.py 128 > gpurun_out/r06_c2_$P.log 2>&1
  DB=$(find gpurun_out/p_$P -name "*.db" | head -1)
  python tools/rocpd_gaps.py $DB clip_adam 8 > gpurun_out/r06_c2_gaps_$P.txt
  python tools/rocpd_bygrid.py $DB "" 22 > gpurun_out/r06_c2_bygrid_$P.txt
  tail -1 gpurun_out/r06_c2_$P.log; head -3 gpurun_out/r06_c2_gaps_$P.txt
  rm -rf gpurun_out/p_$P
done

#!/bin/bash
# round 6, call 3: replay lanes on streams known to run side by side (the engine's side stream): step time, trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in 2 4; do N_LANES=$L PHASE=replay timeout 300 python tools/dev/replay_trace.py 128 2>&1 | tail -1; done | tee gpurun_out/r06_c3_lanes.txt
OWN_LANES=0 N_LANES=4 PHASE=replay timeout 300 python tools/dev/replay_trace.py 128 2>&1 | tail -1 | tee -a gpurun_out/r06_c3_lanes.txt
PHASE=eager timeout 300 python tools/dev/replay_trace.py 128 2>&1 | tail -1 | tee -a gpurun_out/r06_c3_lanes.txt
for B in 64 32; do for P in eager replay; do N_LANES=2 PHASE=$P timeout 300 python tools/dev/replay_trace.py $B 2>&1 | tail -1; done; done | tee -a gpurun_out/r06_c3_lanes.txt
for P in replay; do
  N_LANES=2 PHASE=$P rocprofv3 --kernel-trace -d gpurun_out/p_$P -o t -- python tools/dev/replay_trace.py 128 > gpurun_out/r06_c3_$P.log 2>&1
  DB=$(find gpurun_out/p_$P -name "*.db" | head -1)
  python tools/rocpd_gaps.py $DB clip_adam 8 > gpurun_out/r06_c3_gaps_$P.txt
  python tools/rocpd_bygrid.py $DB "" 22 > gpurun_out/r06_c3_bygrid_$P.txt
  tail -1 gpurun_out/r06_c3_$P.log; head -12 gpurun_out/r06_c3_gaps_$P.txt
  rm -rf gpurun_out/p_$P
done

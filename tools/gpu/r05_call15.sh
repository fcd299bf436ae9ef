#!/bin/bash
# round 5, call 15: the final tree's step in launch order (both queues) and the STN head's un-profiled cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python tools/dev/stn_time.py 128 2>&1 | tail -1 | tee gpurun_out/r05_c15_stn.txt
rocprofv3 --kernel-trace -d gpurun_out/p_r05_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_r05_seq.log 2>&1
DB=$(find gpurun_out/p_r05_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r05_step_sequence.txt; wc -l gpurun_out/r05_step_sequence.txt
rm -rf gpurun_out/p_r05_seq

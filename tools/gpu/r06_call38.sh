#!/bin/bash
# round 6, call 38: remaining focus tests; kernel table + idle gaps of the recorded tfl step at the README's batch 16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_text_focus.py tests/test_gpu_replay.py -m gpu -x -q -k "padding or focus_step" 2>&1 | tail -5
S=10; W=6
rocprofv3 --kernel-trace -d gpurun_out/p_t -o t -- python bench.py --config tfl --batch 16 --steps $S --warmup $W --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_t -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06e_tfl_b16_bygrid.txt; head -40 gpurun_out/r06e_tfl_b16_bygrid.txt; tail -1 gpurun_out/r06e_tfl_b16_bygrid.txt
ls tools/*.py | head -20
rm -rf gpurun_out/p_t

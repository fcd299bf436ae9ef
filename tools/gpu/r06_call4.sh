#!/bin/bash
# round 6, call 4: recorded step in the engine: replay tests, then every multi-step model test (they now replay from step 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q 2>&1 | tail -15 | tee gpurun_out/r06_c4_replay_tests.txt
true
timeout 600 python -m pytest tests/test_text_focus.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_c4_tfl_tests.txt

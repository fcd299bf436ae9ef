#!/bin/bash
# round 6, call 5: replay tests + TextFocusLoss engine gradient test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py -q 2>&1 | tail -25 | tee gpurun_out/r06_c5_replay_tests.txt
timeout 600 python -m pytest tests/test_text_focus.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_c5_tfl_tests.txt

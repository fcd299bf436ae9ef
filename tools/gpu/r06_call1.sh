#!/bin/bash
# round 6, call 1: step replay feasibility (csrc/replay.hip): capture the c3 step, re-issue it from C; eager vs replay vs hipGraphLaunch
# host / step times at B = 128 and 64; the new TextFocusLoss engine-gradient test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export OUT=gpurun_out
timeout 600 python tools/dev/replay_probe.py 128 2>&1 | tail -25 | tee gpurun_out/r06_c1_replay_b128.txt
cp gpurun_out/replay_nodes.txt gpurun_out/r06_c1_replay_nodes_b128.txt
GRAPH_LAUNCH=0 timeout 600 python tools/dev/replay_probe.py 64 2>&1 | tail -12 | tee gpurun_out/r06_c1_replay_b64.txt
timeout 900 python -m pytest tests/test_text_focus.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06_c1_tfl_tests.txt

#!/bin/bash
# round 6, call 39: focus-loss padding / recorded-step tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_text_focus.py tests/test_gpu_replay.py -m gpu -q -k "padding or focus_step or masked" 2>&1 | tail -25

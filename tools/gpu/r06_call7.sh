#!/bin/bash
# round 6, call 7: does mode 3 train like mode 1?  300 steps, B = 128 (tools/mode3_equivalence.py); replay tests again
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/mode3_equivalence.py --steps 300 --batch 128 --out gpurun_out/r06_mode3_equivalence.json 2>&1 | grep -v amdgpu.ids | tail -75
timeout 600 python -m pytest tests/test_gpu_replay.py -q 2>&1 | tail -3

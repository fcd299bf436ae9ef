#!/bin/bash
# round 6, call 49: masked full-size step vs oracle
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/test_margins.txt
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "oracle_mask" 2>&1 | tail -8; cat gpurun_out/test_margins.txt 2>/dev/null | tail -3

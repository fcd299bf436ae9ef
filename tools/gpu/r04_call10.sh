#!/bin/bash
# fp64 arbitration of the element-wise gradient gate (margins are written to gpurun_out/test_margins.txt by the test)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/test_margins.txt
timeout 1200 python -m pytest tests/test_gpu_models.py -q -m gpu -k "gradients_elementwise" 2>&1 | tail -15
cat gpurun_out/test_margins.txt

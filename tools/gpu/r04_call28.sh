#!/bin/bash
# conv3x3_c64_wgrad_kernel with the priming loads issued together: timing + parity
cd $GRAFT_REPO_ROOT
timeout 120 python tools/kbench.py conv 2>&1 | grep -E "srb 3x3|up 3x3"
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad or conv2d or conv3x3 or halo" 2>&1 | tail -4

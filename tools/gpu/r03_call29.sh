#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -m gpu 2>&1 > gpurun_out/c29_tests.log; grep -E "^FAILED|^E  |passed|failed" gpurun_out/c29_tests.log | head -20

#!/bin/bash
# profile set r04a: the step with the single-pass attention backward (kernel trace, by-grid, gaps, FETCH / WRITE / MFMA PMC)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "single_pass" 2>&1 | tail -3
bash tools/profile_round.sh r04a

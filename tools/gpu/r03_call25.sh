#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FOCR_LIB=$PWD/fudanocr_amd/libfocr_hip_noslp.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -60 > gpurun_out/c25_noslp_tests.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -60 > gpurun_out/c25_new_tests.log
tail -3 gpurun_out/c25_noslp_tests.log gpurun_out/c25_new_tests.log

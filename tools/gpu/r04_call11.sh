#!/bin/bash
# FeatureEnhancer chains with next-tile prefetch vs the round-3 kernels (standalone; each build checks itself against the
# one-thread-per-row fp64 chains)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== base" | tee gpurun_out/r04_c11_fe.log
timeout 200 build/fe_ubench_base 128 2>&1 | grep -E "FAIL|us " | tee -a gpurun_out/r04_c11_fe.log
echo "== prefetch" | tee -a gpurun_out/r04_c11_fe.log
timeout 200 build/fe_ubench 128 2>&1 | grep -E "FAIL|ok|us " | grep -v "  ok$" | tee -a gpurun_out/r04_c11_fe.log
timeout 200 build/fe_ubench 128 2>&1 | grep -c "ok$" | tee -a gpurun_out/r04_c11_fe.log
echo "== base again" | tee -a gpurun_out/r04_c11_fe.log
timeout 200 build/fe_ubench_base 128 2>&1 | grep -E "us " | tee -a gpurun_out/r04_c11_fe.log

#!/bin/bash
# round 3, call 1: fused FeatureEnhancer chains -- ubench correctness/timing, new tests, model parity, bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/c1
( timeout 300 build/fe_ubench 128 ) > ${O}_ubench.log 2>&1
echo "ubench rc=$?" >> ${O}_ubench.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "feature_enhancer" > ${O}_t_new.log 2>&1
echo "rc=$?" >> ${O}_t_new.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu > ${O}_t_models.log 2>&1
echo "rc=$?" >> ${O}_t_models.log
timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 20 > ${O}_b_c3.log 2>&1
FOCR_FE_FUSED=0 timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 20 > ${O}_b_c3_unfused.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --config c2 --steps 40 --warmup 20 > ${O}_b_c2.log 2>&1
timeout 300 python tools/dev/host_profile.py > ${O}_host.log 2>&1
tail -5 ${O}_ubench.log ${O}_t_new.log ${O}_t_models.log
tail -c 600 ${O}_b_c3.log; tail -c 300 ${O}_b_c3_unfused.log; tail -c 300 ${O}_b_c2.log

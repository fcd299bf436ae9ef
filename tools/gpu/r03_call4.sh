#!/bin/bash
# round 3, call 4: fused concat-PE + QKV projection, D in the backward chain, BN fold; targeted tests + bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/c4
rm -f gpurun_out/test_margins.txt
( timeout 300 build/fe_ubench 128 ) > ${O}_ubench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py tests/test_sld.py -q -m gpu -k "feature_enhancer or attention or batchnorm or golden or oracle or full_size" > ${O}_pytest.log 2>&1
echo "rc=$?" >> ${O}_pytest.log
timeout 300 python bench.py --no-cpu-baseline --steps 60 > ${O}_b_c3.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 60 --config c2 > ${O}_b_c2.log 2>&1
tail -22 ${O}_ubench.log; tail -6 ${O}_pytest.log; tail -c 300 ${O}_b_c3.log; tail -c 200 ${O}_b_c2.log

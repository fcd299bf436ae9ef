#!/bin/bash
# round 6, call 35: four-channel (--mask) forms of the two 9x9 layers: kernel tests, the mask golden tests, masked c3 step + kernel table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "test_conv2d" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "mask" 2>&1 | tail -5
python bench.py --config c3 --mask --steps 40 --warmup 10 --no-other-configs > gpurun_out/c3_mask.log 2>&1; tail -1 gpurun_out/c3_mask.log | cut -c1-260
S=8; W=6
rocprofv3 --kernel-trace -d gpurun_out/p_m -o t -- python bench.py --config c3 --mask --steps $S --warmup $W --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_m -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06e_c3_mask_bygrid.txt; grep -n "conv9x9\|conv_fwd_kernel\|conv_wgrad_kernel\|conv_wgrad_bx3_kernel" gpurun_out/r06e_c3_mask_bygrid.txt | head
rm -rf gpurun_out/p_m

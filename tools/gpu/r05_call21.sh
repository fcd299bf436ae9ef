#!/bin/bash
# round 5, call 21: tps_bwd with one pixel per thread: TPS / golden tests, kernel time in the step trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "tps" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "train_mse_golden or traj3 or e2e_ctc_golden" 2>&1 | tail -1
rocprofv3 --kernel-trace -d gpurun_out/p_r05_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_r05_seq.log 2>&1
DB=$(find gpurun_out/p_r05_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r05_step_sequence.txt; wc -l gpurun_out/r05_step_sequence.txt
grep "tps_\|weight_flip" gpurun_out/r05_step_sequence.txt | cut -c1-110
rm -rf gpurun_out/p_r05_seq

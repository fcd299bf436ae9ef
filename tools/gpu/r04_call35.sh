#!/bin/bash
# focr_zero for zero_grad + tiny-Cin 3x3 weight gradient (STN conv1): parity, step sequence tail, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "test_conv2d or clip_adam" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "train_mse_golden or traj3 or stn" 2>&1 | tail -2
rocprofv3 --kernel-trace -d gpurun_out/p_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_seq.log 2>&1
DB=$(find gpurun_out/p_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r04_step_sequence.txt; rm -rf gpurun_out/p_seq
head -3 gpurun_out/r04_step_sequence.txt | cut -c1-120; tail -8 gpurun_out/r04_step_sequence.txt | cut -c1-120
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done

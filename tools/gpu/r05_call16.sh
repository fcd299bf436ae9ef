#!/bin/bash
# round 5, call 16: tiny 1x1 VALU kernel for the STN head's fc2 data gradient (60 us on the tiled fp32 kernel): linear /
# conv tests, STN golden tests, STN head timing, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c16
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear or conv2d" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "train_mse_golden or traj3 or eval_golden" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 120 python tools/dev/stn_time.py 128 2>&1 | tail -1
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done

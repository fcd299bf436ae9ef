#!/bin/bash
# BatchNorm backward with the ticketed group fold (2 launches instead of 3): parity incl. ring cycling, golden models, step A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batchnorm or bn" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "full_size or train_mse_golden or traj3" 2>&1 | tail -3
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  for v in "tickets FOO=1" "notickets FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_notickets.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

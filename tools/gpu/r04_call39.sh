#!/bin/bash
# second BatchNorm of the residual block normalised on the packed projection's load (FOCR_BN2_FUSE) + host-side CTC offsets:
# parity, interleaved step A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "srb or feature_enhancer or ctc or batchnorm" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "golden or elementwise or traj_fixed" 2>&1 | tail -3
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  for v in "fused FOCR_BN2_FUSE=1" "apart FOCR_BN2_FUSE=0"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

#!/bin/bash
# small-tensor BatchNorm (one launch forward, one backward; STN head layers): parity + STN timing + interleaved step A/B
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batchnorm or bn" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "train_mse_golden or e2e_ctc_golden or traj3" 2>&1 | tail -3
python tools/dev/stn_time.py 128 | tail -1
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_nosmallbn.so python tools/dev/stn_time.py 128 | tail -1
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in "small FOO=1" "nosmall FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_nosmallbn.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

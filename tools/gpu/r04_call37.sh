#!/bin/bash
# persistent LSTM: nontemporal stores for the fp32 outputs (no dirty lines for the per-step release to flush): kernel time, parity, step
cd $GRAFT_REPO_ROOT
for v in "plain FOO=1" "nt FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lstmnt.so"; do
  set -- $v
  echo "== $1"; env $2 timeout 120 python tools/kbench.py lstm 2>&1 | grep "lstm"
done
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lstmnt.so timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lstm" 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in "plain FOO=1" "nt FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lstmnt.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

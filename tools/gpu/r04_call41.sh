#!/bin/bash
# vectorised maxpool forward (four channels per thread): parity + interleaved step A/B
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "maxpool" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "crnn or e2e_ctc_golden" 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  for v in "vec4 FOO=1" "scalar FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_novec4.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

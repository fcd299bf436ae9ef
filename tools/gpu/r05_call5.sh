#!/bin/bash
# round 5, call 5: attention forward with the relative-max softmax (tuning key 4 = 2) -- micro-benchmark incl. adversarial
# score ranges, attention + FeatureEnhancer tests and goldens on it, step A/B; persistent-LSTM phase stamps (LP_TRACE build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c5
( timeout 200 build/attn_ubench 128 1 1 fm ) > ${O}_fm.log 2>&1; cat ${O}_fm.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention or feature" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "golden or fresh_batch" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in 2 1; do
    ms=$(timeout 300 $B --tuning 4=$v 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r forward variant $v: $ms"
  done
done
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lptrace.so timeout 200 python tools/dev/lstm_phases.py 128 > ${O}_lstm_phases.txt 2>&1; cat ${O}_lstm_phases.txt
timeout 120 python tools/dev/lstm_bench.py 128 2>&1 | tail -6

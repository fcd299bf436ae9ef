#!/bin/bash
# round 5, call 3: streaming linear weight gradient for Cout = 64 (the FeatureEnhancer's 128 -> 64 projection): standalone
# correctness against a double-precision reference + timing, the FeatureEnhancer / linear tests, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c3
( timeout 200 build/lwgrad_ubench ) > ${O}_lw.log 2>&1; grep -A4 -E "^proj|^out" ${O}_lw.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "linear or feature or wgrad or fe_" > ${O}_pytest.log 2>&1; tail -3 ${O}_pytest.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "train_mse_golden or elementwise_vs_oracle or feature_enhancer" > ${O}_pytest2.log 2>&1; tail -3 ${O}_pytest2.log
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in 1 0; do
    ms=$(FOCR_LW_HALF=$v timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r 128->64 weight gradient on the streaming kernel=$v: $ms"
  done
done

#!/bin/bash
# round 5, call 11: backward FeatureEnhancer chains with the 4 x 4 register-transposed weight staging (conflict-free 8-byte
# LDS stores), two-phase W staging of the LSTM scans and the 9x9 output layer: correctness, timing, tests, step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c11
for b in 2 128; do echo "== B=$b"; timeout 200 build/fe_ubench $b 2>&1 | grep -vE "^fe_ubench"; done > ${O}_fe.log 2>&1; grep -E "^==|FAIL|^fe_" ${O}_fe.log | head -40; grep -c " ok" ${O}_fe.log
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "feature or fe_ or linear or qkv or lstm or crnn or conv9x9 or small_cout or 9x9" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "golden or elementwise_vs_oracle or fresh_batch" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 120 python tools/dev/lstm_bench.py 128 2>&1 | head -2
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done

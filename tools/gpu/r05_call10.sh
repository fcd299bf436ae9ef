#!/bin/bash
# round 5, call 10: FeatureEnhancer chains with the two-phase weight staging: standalone correctness (fp64 row chains) and
# timing at 2 048 / 131 072 rows, kernel + model tests, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c10
for b in 2 128; do echo "== B=$b"; timeout 200 build/fe_ubench $b 2>&1 | grep -vE "^fe_ubench"; done > ${O}_fe.log 2>&1; grep -E "^==|FAIL|^fe_|max" ${O}_fe.log | head -60
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "feature or fe_ or linear or qkv" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "golden or elementwise_vs_oracle or fresh_batch" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done
for c in c2 c1; do timeout 300 python bench.py --config $c --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$c', d['ms_per_step'], d['value'])"; done

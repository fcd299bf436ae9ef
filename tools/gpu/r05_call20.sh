#!/bin/bash
# round 5, call 20: c1 (TSRN) re-measured (16.8 ms in r05_call19's 20-step subprocess run against 13.8 before); early weight
# flip on the side stream: trajectory / golden tests, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for r in 1 2; do
  for c in c1 c2; do timeout 300 python bench.py --config $c --steps 40 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$c', d['ms_per_step'], d['value'])"; done
done
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -q -m gpu -k "traj or golden or dp_engine or fresh_batch" 2>&1 | tail -2
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done

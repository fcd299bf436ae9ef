#!/bin/bash
# round 5, call 24: why two gloo ranks on one device take 830 ms per step (batch 32 each): LSTM launch forms, finite loss
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['value'], d['final_loss'], d.get('exposed_comm_ms'))"; }
timeout 300 python bench.py --steps 20 --warmup 10 --batch 32 --no-cpu-baseline --no-other-configs 2>/dev/null | show "1 rank B=32"
for t in 1 2 0; do
  FOCR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 5 --batch 32 --tuning 2=$t 2>/dev/null | show "2 gloo ranks, LSTM tuning $t"
done
FOCR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 5 --batch 32 --config c2 2>/dev/null | show "2 gloo ranks, c2 (no recognizer)"

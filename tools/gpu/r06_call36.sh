#!/bin/bash
# round 6, call 36: batch sweep (the reference's README runs --batch_size=16, its yaml default is 512): img/s against batch for c3, c1, tfl
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c3 c1 tfl; do
for B in 8 16 32 48 64 100 128 256 512; do
  S=30; [ $B -ge 256 ] && S=12; [ $C = tfl ] && S=10
  timeout 600 python bench.py --config $C --batch $B --steps $S --warmup 8 --no-cpu-baseline --no-other-configs > gpurun_out/sweep.log 2>&1
  python - "$C" "$B" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/sweep.log").read().strip().splitlines()[-1])
    print("%s B=%4s  %8.3f ms/step  %9.1f img/s  recorded=%s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], bool(d["config"].get("recorded_step"))))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open("gpurun_out/sweep.log").read()[-1500:])
PY
done; done 2>&1 | tee gpurun_out/r06_batch_sweep.txt

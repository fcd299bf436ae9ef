#!/bin/bash
# round 6, call 42: small batches: one queue vs two (cross-queue event waits cost more than the overlap buys when kernels are short?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in c3 tfl c1; do for B in 8 16 32 64; do for SIDE in 1 0; do
  FOCR_WGRAD_SIDE=$SIDE timeout 600 python bench.py --config $C --batch $B --steps 30 --warmup 8 --no-cpu-baseline --no-other-configs > gpurun_out/sweep.log 2>&1
  python - "$C" "$B" "$SIDE" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/sweep.log").read().strip().splitlines()[-1])
    print("%s B=%4s side=%s  %8.3f ms/step  %9.1f img/s  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"], d["value"], {k: d["config"]["recorded_step"][k] for k in ("nodes", "lanes", "waits")}))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open("gpurun_out/sweep.log").read()[-1500:])
PY
done; done; done 2>&1 | tee gpurun_out/r06_small_batch_queues.txt

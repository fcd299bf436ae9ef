#!/bin/bash
# round 6, call 41: label capacity bucket of the recorded focus steps (padding costs decoder positions; finer buckets cost recordings)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for C in tfl sfl; do for LB in 4 8 16 32; do for B in 16 128; do
  FOCR_LABEL_BUCKET=$LB timeout 600 python bench.py --config $C --batch $B --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs > gpurun_out/sweep.log 2>&1
  python - "$C" "$B" "$LB" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/sweep.log").read().strip().splitlines()[-1])
    print("%s bucket %2s B=%4s  %8.3f ms/step  %9.1f img/s  recorded=%s" % (sys.argv[1], sys.argv[3], sys.argv[2], d["ms_per_step"], d["value"], bool(d["config"].get("recorded_step"))))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open("gpurun_out/sweep.log").read()[-1500:])
PY
done; done; done 2>&1 | tee gpurun_out/r06_label_bucket_sweep.txt

#!/bin/bash
# round 6, call 37: recordable focus-loss step (padded labels): tests, then tfl / sfl at the README's batch 16 and at 128
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_text_focus.py tests/test_gpu_replay.py -m gpu -x -q 2>&1 | tail -15
for C in tfl sfl; do for B in 16 64 128; do
  timeout 600 python bench.py --config $C --batch $B --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs > gpurun_out/sweep.log 2>&1
  python - "$C" "$B" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/sweep.log").read().strip().splitlines()[-1])
    print("%s B=%4s  %8.3f ms/step  %9.1f img/s  recorded=%s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], d["config"].get("recorded_step")))
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e); print(open("gpurun_out/sweep.log").read()[-2500:])
PY
done; done 2>&1 | tee gpurun_out/r06_focus_replay_bench.txt

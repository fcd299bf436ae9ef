#!/bin/bash
# c5 (SLD recognizer) regression check: small-BatchNorm path with the element cap vs without the path; c3 re-check
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "small FOO=1" "nosmall FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_nosmallbn.so"; do
    set -- $v
    for c in c5 c3 c1; do
      ms=$(env $2 timeout 300 python bench.py --config $c --steps 30 --warmup 15 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])")
      echo "round $r $1 $c: $ms"
    done
  done
done

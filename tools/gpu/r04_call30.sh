#!/bin/bash
# weight gradients of the last k residual blocks parked for the STN head's backward (FOCR_PARK_TAIL=k) + BatchNorm backward
# slab caps: interleaved step A/B
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in "tail0 FOCR_PARK_TAIL=0" "tail1 FOCR_PARK_TAIL=1" "tail2 FOCR_PARK_TAIL=2" "tail3 FOCR_PARK_TAIL=3" "slabs1024 FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_slabs1024.so" "slabs512 FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_slabs512.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

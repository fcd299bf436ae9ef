#!/bin/bash
# round 5, call 2: keep-word schedule of the attention forward (tuning key 4) in the standalone micro-benchmark -- fp32 and
# pre-split operands, packed layouts --, the attention tests on the new default, the two-legged CPU baseline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c2
( timeout 200 build/attn_ubench 128 1 1 fm ) > ${O}_fm.log 2>&1; cat ${O}_fm.log
( FOCR_UB_MASKV=1 timeout 300 build/attn_ubench 128 ) > ${O}_ub_mv1.log 2>&1; grep -E "^fwd|PACKED|PLANES" -A1 ${O}_ub_mv1.log | grep -v "^--" | cut -c1-230
( FOCR_UB_MASKV=0 timeout 300 build/attn_ubench 128 ) > ${O}_ub_mv0.log 2>&1; grep -E "^fwd|PACKED p=0.1: fwd" ${O}_ub_mv0.log | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "attention" > ${O}_pytest.log 2>&1; tail -3 ${O}_pytest.log
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in 1 0; do
    ms=$(timeout 300 $B --tuning 4=$v 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r mask-schedule $v: $ms"
  done
done
timeout 200 python -c "import bench, json; print(json.dumps(bench.cpu_baseline_guarded('c3')))" | tail -1

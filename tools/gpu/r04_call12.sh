#!/bin/bash
# side stream: fewer row splits of the streaming linear weight gradient (512 -> 256 -> 128 slots), interleaved A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c12.log
run() { FOCR_LIB=$2 timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$1', r['ms_per_step'], r['value'])" | tee -a gpurun_out/r04_c12.log; }
for rep in 1 2; do
  run lw512 fudanocr_amd/libfocr_hip.so
  run lw256 $PWD/fudanocr_amd/libfocr_hip_lw256.so
  run lw128 $PWD/fudanocr_amd/libfocr_hip_lw128.so
done

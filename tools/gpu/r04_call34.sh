#!/bin/bash
# small-tensor BatchNorm with eight rows in flight: parity, STN timing, step sequence (kernel durations), step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "batchnorm or bn" 2>&1 | tail -2
python tools/dev/stn_time.py 128 | tail -1
rocprofv3 --kernel-trace -d gpurun_out/p_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_seq.log 2>&1
DB=$(find gpurun_out/p_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r04_step_sequence.txt; rm -rf gpurun_out/p_seq
grep bn_small gpurun_out/r04_step_sequence.txt | awk '{print $2, $5, $6}'
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in "small FOO=1" "nosmall FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_nosmallbn.so"; do
    set -- $v
    ms=$(env $2 timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $1: $ms"
  done
done

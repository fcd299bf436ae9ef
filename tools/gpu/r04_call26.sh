#!/bin/bash
# fused first recognizer layer (conv0 + relu + pooling0, csrc/crnn_conv0_pool.hip): parity + interleaved step A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv0 or maxpool" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "crnn" 2>&1 | tail -5
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in default conv0off; do
    case $v in
      default) env="FOCR_CONV0_POOL=1";;
      conv0off) env="FOCR_CONV0_POOL=0";;
    esac
    ms=$(env $env timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
    echo "round $r $v: $ms"
  done
done

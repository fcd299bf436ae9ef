#!/bin/bash
# round 6, call 50: relu backward in the data-gradient epilogue of the frozen recognizers' ResNet blocks: tests, tfl / sfl A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "halo" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_text_focus.py tests/test_gpu_replay.py -m gpu -x -q -k "focus" 2>&1 | tail -3
for C in tfl sfl; do for B in 16 128; do for M in 0 1 0 1; do
  FOCR_HALO_MASK=$M timeout 600 python bench.py --config $C --batch $B --steps 20 --warmup 8 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$C B=$B FOCR_HALO_MASK=$M', d['ms_per_step'], d['config']['recorded_step']['nodes'])"
done; done; done | tee gpurun_out/r06_halo_mask_ab.txt

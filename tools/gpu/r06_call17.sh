#!/bin/bash
# round 6, call 17: halo convolution tile shapes for narrow maps (8 x 16, 16 x 8): C-ABI test, conv tests, c5 / tfl / sfl / c3 timings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "halo or conv2d or conv" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_sld.py tests/test_text_focus.py -q -m gpu 2>&1 | tail -4
for C in c5 tfl sfl c3 c1; do timeout 600 python bench.py --config $C --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C', d['ms_per_step'], d['value'], d['roofline']['step_algorithmic_tflops'])"
done | tee gpurun_out/r06_c17_shapes.txt

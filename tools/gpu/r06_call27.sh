#!/bin/bash
# round 6, call 27: small attention (16 heads x 64) with one block per (batch, head) over all query rows: tests, tfl / sfl timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_text_focus.py tests/test_sld.py -q -m gpu 2>&1 | tail -4
for C in tfl sfl; do timeout 600 python bench.py --config $C --steps 20 --warmup 6 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C', d['ms_per_step'], d['value'], d['roofline']['step_algorithmic_tflops'])"
done | tee gpurun_out/r06_c27.txt

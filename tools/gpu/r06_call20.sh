#!/bin/bash
# round 6, call 20: eval-mode BatchNorm folded into the frozen recognizers' convolutions (text- / stroke-focus losses): tests, tfl / sfl A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_text_focus.py tests/test_sld.py -q -m gpu 2>&1 | tail -4
for F in 1 0; do for C in tfl sfl; do FOCR_FOLD_BN=$F timeout 600 python bench.py --config $C --steps 20 --warmup 6 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$C fold=$F', d['ms_per_step'], d['value'], d['roofline']['step_algorithmic_tflops'])"
done; done | tee gpurun_out/r06_c20_fold.txt

#!/bin/bash
# round 6, call 8: N1 measured: full-size tfl / sfl step vs oracle, bench --config tfl / sfl (B sweep), kernel breakdown of tfl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_text_focus.py -q -m gpu -k "full_size" 2>&1 | tail -8 | tee gpurun_out/r06_c8_tfl_fullsize.txt
for C in tfl sfl; do for B in 64 128; do
timeout 600 python bench.py --config $C --batch $B --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$C B=$B', d['ms_per_step'], d['value'], r['kernel'][:40], r['frac'], r['avg_launch_ms'], r['step_algorithmic_tflops'])"
done; done | tee gpurun_out/r06_c8_tfl_bench.txt
rocprofv3 --kernel-trace -d gpurun_out/p_tfl -o t -- python bench.py --config tfl --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/r06_c8_tfl_prof.log 2>&1
DB=$(find gpurun_out/p_tfl -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" 10 > gpurun_out/r06_tfl_bygrid.txt; head -40 gpurun_out/r06_tfl_bygrid.txt
python tools/rocpd_gaps.py $DB clip_adam 4 > gpurun_out/r06_tfl_gaps.txt; head -3 gpurun_out/r06_tfl_gaps.txt
rm -rf gpurun_out/p_tfl

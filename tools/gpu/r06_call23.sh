#!/bin/bash
# round 6, call 23: c1 per-queue busy time and one-queue timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for S in 1 0; do FOCR_WGRAD_SIDE=$S timeout 600 python bench.py --config c1 --steps 30 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('c1 side=$S', d['ms_per_step'], d['value'])"
done | tee gpurun_out/r06_c23.txt
rocprofv3 --kernel-trace -d gpurun_out/p_c1 -o t -- python bench.py --config c1 --steps 8 --warmup 6 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_c1 -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 3 > gpurun_out/r06_c1_sequence.txt
python tools/rocpd_bygrid.py $DB "" 16 > gpurun_out/r06c_c1_bygrid.txt
rm -rf gpurun_out/p_c1
python - <<'PY'
rows=[l.split(None,5) for l in open('gpurun_out/r06_c1_sequence.txt')]
import collections
byq=collections.defaultdict(list)
for r in rows: byq[r[3]].append((float(r[0]),float(r[1]),r[5][:50]))
for q,v in byq.items():
    print(q, len(v), 'kernels, busy %.0f us, first %.0f last end %.0f'%(sum(d for _,d,_ in v), v[0][0], max(s+d for s,d,_ in v)))
PY
head -24 gpurun_out/r06c_c1_bygrid.txt

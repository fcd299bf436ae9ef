#!/bin/bash
# round 6, call 46: c3 at batch 16 (README batch): kernel table, which kernels do not scale down with the batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
S=20; W=6
rocprofv3 --kernel-trace -d gpurun_out/p_t -o t -- python bench.py --config c3 --batch 16 --steps $S --warmup $W --no-cpu-baseline --no-other-configs > /dev/null 2>&1
DB=$(find gpurun_out/p_t -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "" $((S+W)) > gpurun_out/r06e_c3_b16_bygrid.txt; head -60 gpurun_out/r06e_c3_b16_bygrid.txt
python tools/rocpd_gaps.py $DB $((S+W)) 2>/dev/null | head -12
rm -rf gpurun_out/p_t

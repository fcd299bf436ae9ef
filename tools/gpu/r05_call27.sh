#!/bin/bash
# round 5, call 27: what the ~10 ms floor of the step is made of: kernel-busy union against the step span at batch 8 and 64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for b in 8 64; do
  rocprofv3 --kernel-trace -d gpurun_out/p_r05_b$b -o bench -- python bench.py --batch $b --steps 12 --warmup 6 --no-cpu-baseline --no-other-configs > gpurun_out/p_r05_b$b.log 2>&1
  DB=$(find gpurun_out/p_r05_b$b -name "*.db" | head -1)
  echo "B=$b: $(python tools/rocpd_gaps.py $DB clip_adam 8 | head -1)"
  rm -rf gpurun_out/p_r05_b$b
done | tee gpurun_out/r05_c27_floor.txt

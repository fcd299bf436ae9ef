#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 0; do
  FOCR_ATTN_PLANES=$v rocprofv3 --kernel-trace -d gpurun_out/p_c12_$v -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline > gpurun_out/c12_kt_$v.log 2>&1
  DB=$(find gpurun_out/p_c12_$v -name "*.db" | head -1)
  python tools/rocpd_stats.py $DB gpurun_out/c12_stats_$v.csv 2> /dev/null
  rm -rf gpurun_out/p_c12_$v
done
for v in 1 0; do echo "== planes=$v"; grep -E "attn_|fe_qkv_fwd|fe_bwd_b|fe_bwd_qkv|fe_fwd_a|linear_wgrad" gpurun_out/c12_stats_$v.csv | cut -d, -f1-4 | sed 's/_Z[0-9]*//' | cut -c1-90; done

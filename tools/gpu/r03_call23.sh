#!/bin/bash
# attention kernels: pairwise split with scalar subtractions, with / without the SLP vectoriser (v_pk_* f32), vs the
# previous source (base) -- interleaved twice to see the run-to-run spread
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for v in base base_noslp new new_noslp; do
  b=build/attn_ubench_$v; [ $v = new ] && b=build/attn_ubench; [ $v = new_noslp ] && b=build/attn_ubench_noslp
  echo "== $v"; timeout 300 $b 128 2>&1 | grep -E "^fwd|^bwd" | sed 's/\[mask.*//; s/max diff.*//'
done; done | tee gpurun_out/c23_attn.log

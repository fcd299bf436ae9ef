#!/bin/bash
# conv3x3_c64_wgrad_kernel: where its 43 us go (timing-only ablation builds, tools/kbench.py conv)
cd $GRAFT_REPO_ROOT
for v in "" _c3w_mfma _c3w_stage _c3w_load _c3w_epi; do
  echo "== libfocr_hip$v"
  FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip$v.so timeout 120 python tools/kbench.py conv 2>&1 | grep -E "srb 3x3|up 3x3"
done

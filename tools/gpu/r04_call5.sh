#!/bin/bash
# SQ counters of the single-pass attention backward vs the two-pass kernels (standalone ubench, B = 128)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o p -- $R/build/attn_ubench_b1 128 1 2 b1 > /tmp/pmc_$n.log 2>&1
  python $R/tools/rocpd_pmc.py $(find /tmp/pmc_$n -name "*.db" | head -1) $R/gpurun_out/r04_pmc_bwd1_$n.csv
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM
run c SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE
grep -h "attn_bwd" $R/gpurun_out/r04_pmc_bwd1_*.csv

cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o p -- python /root/repo/tools/kbench.py attn > /tmp/pmc_$n.log 2>&1
  python /root/repo/tools/rocpd_pmc.py $(find /tmp/pmc_$n -name "*.db" | head -1) /root/repo/gpurun_out/r01j_pmc_attn_$n.csv
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM
run c SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_EXP_GDS
grep -h "attn_fwd_bx3\|attn_bwd" /root/repo/gpurun_out/r01j_pmc_attn_*.csv | head -80

#!/bin/bash
# One gpurun call: kernel trace + PMC passes of the bench command (separate passes, --kernel-trace only), distilled
# into gpurun_out/ (copy what should be judged into profiles/).  usage: bash tools/profile_round.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline $@"
rocprofv3 --kernel-trace -d gpurun_out/p_${TAG}_kt -o bench -- $B > gpurun_out/p_${TAG}_kt.log 2>&1
DB=$(find gpurun_out/p_${TAG}_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/${TAG}_bench_kernel_stats.csv 2> gpurun_out/${TAG}_kt_total.txt
python tools/rocpd_bygrid.py $DB conv3x3_halo 13 > gpurun_out/${TAG}_halo_bygrid.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/p_${TAG}_$C -o bench -- $B > gpurun_out/p_${TAG}_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/p_${TAG}_mfma -o bench -- $B > gpurun_out/p_${TAG}_mfma.log 2>&1
F=$(find gpurun_out/p_${TAG}_FETCH_SIZE -name "*.db" | head -1); W=$(find gpurun_out/p_${TAG}_WRITE_SIZE -name "*.db" | head -1)
python tools/pmc_traffic.py $F $W 128 "$TAG: rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- $B" gpurun_out/${TAG}_pmc_traffic.json
python tools/rocpd_pmc.py $F gpurun_out/${TAG}_pmc_FETCH_SIZE.csv; python tools/rocpd_pmc.py $W gpurun_out/${TAG}_pmc_WRITE_SIZE.csv
M=$(find gpurun_out/p_${TAG}_mfma -name "*.db" | head -1); python tools/rocpd_pmc.py $M gpurun_out/${TAG}_pmc_mfma.csv
python tools/rocpd_bygrid.py $DB "" 13 > gpurun_out/${TAG}_all_bygrid.txt; python tools/rocpd_gaps.py $DB clip_adam 6 > gpurun_out/${TAG}_gaps.txt; python tools/pmc_mfma_util.py gpurun_out/${TAG}_pmc_mfma.csv > gpurun_out/${TAG}_mfma_util.txt 2>&1; rm -rf gpurun_out/p_${TAG}_kt gpurun_out/p_${TAG}_FETCH_SIZE gpurun_out/p_${TAG}_WRITE_SIZE gpurun_out/p_${TAG}_mfma
head -12 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-120; cat gpurun_out/${TAG}_pmc_traffic.json | head -30

#!/bin/bash
# kernel trace with the weight gradients IN LINE (FOCR_WGRAD_SIDE=0): every kernel at its un-overlapped in-step cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline"
FOCR_WGRAD_SIDE=0 rocprofv3 --kernel-trace -d gpurun_out/p_serial_kt -o bench -- $B > gpurun_out/p_serial_kt.log 2>&1
DB=$(find gpurun_out/p_serial_kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB gpurun_out/r04b_serial_kernel_stats.csv 2> gpurun_out/r04b_serial_kt_total.txt
python tools/rocpd_bygrid.py $DB "" 13 > gpurun_out/r04b_serial_bygrid.txt
rm -rf gpurun_out/p_serial_kt; cat gpurun_out/r04b_serial_kt_total.txt; head -50 gpurun_out/r04b_serial_bygrid.txt | cut -c1-110

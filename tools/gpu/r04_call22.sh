#!/bin/bash
# LSTM exchange mode 2 (plain stores kept in L2 + drained relaxed counter; relaxed poll; sc1 loads): speed + parity probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/r04_c22.log
export FOCR_LIB=$PWD/fudanocr_amd/libfocr_hip_x2.so
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "lstm" 2>&1 | tail -3 | tee -a gpurun_out/r04_c22.log
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "crnn or e2e or full_size" 2>&1 | tail -3 | tee -a gpurun_out/r04_c22.log
for rep in 1 2; do timeout 300 python bench.py --steps 40 --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('step', r['ms_per_step'], r['value'], r['final_loss'])" | tee -a gpurun_out/r04_c22.log; done
B="python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs"
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/p_l_kt -o bench -- $B > gpurun_out/p_l_kt.log 2>&1
DB=$(find gpurun_out/p_l_kt -name "*.db" | head -1)
python tools/rocpd_bygrid.py $DB "lstm" 13 | cut -c1-110 | tee -a gpurun_out/r04_c22.log
rm -rf gpurun_out/p_l_kt

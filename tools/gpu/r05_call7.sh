#!/bin/bash
# round 5, call 7: persistent LSTM exchange WITHOUT the agent-scope release (LSTM_XCHG = 2 build: payload acknowledged by
# the shared L2 + relaxed counter): correctness on repeated runs, timing, phase stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c7
export FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lx2.so
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or crnn" 2>&1 | tail -1
done
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "crnn or e2e_ctc or decoded or fresh_batch" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 120 python tools/dev/lstm_bench.py 128 2>&1 | tail -6 | tee ${O}_lstm_bench.txt
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lx2trace.so timeout 200 python tools/dev/lstm_phases.py 128 2>/dev/null > ${O}_lstm_phases.txt; cat ${O}_lstm_phases.txt
unset FOCR_LIB
timeout 120 python tools/dev/lstm_bench.py 128 2>&1 | tail -2

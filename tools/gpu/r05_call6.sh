#!/bin/bash
# round 5, call 6: persistent LSTM scans with the W fragments in registers and the partners' rows staged through LDS:
# LSTM / CRNN tests, per-call timing, phase stamps, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c6
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or crnn" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "crnn or e2e_ctc or decoded or fresh_batch" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 120 python tools/dev/lstm_bench.py 128 2>&1 | tail -6 | tee ${O}_lstm_bench.txt
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lptrace.so timeout 200 python tools/dev/lstm_phases.py 128 2>/dev/null > ${O}_lstm_phases.txt; cat ${O}_lstm_phases.txt
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done

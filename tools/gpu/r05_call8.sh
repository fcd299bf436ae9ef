#!/bin/bash
# round 5, call 8: persistent LSTM with the run-time XCD census (releases inside the XCD's L2 when a group's 8 blocks share
# it): LSTM / CRNN tests three times over, the full-size (B = 128) oracle step, timing A/B against the forced agent-scope
# release, phase stamps, step timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c8
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or crnn" > ${O}_pytest_k$i.log 2>&1; tail -1 ${O}_pytest_k$i.log
done
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "crnn or e2e_ctc or decoded or fresh_batch or full_size" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 200 python tools/dev/lstm_bench.py 128 2>&1 | tail -8 | tee ${O}_lstm_bench.txt
FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_lptrace.so timeout 200 python tools/dev/lstm_phases.py 128 2>/dev/null > ${O}_lstm_phases.txt; cat ${O}_lstm_phases.txt
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in 1 2; do
    ms=$(timeout 300 $B --tuning 2=$v 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['final_loss'])")
    echo "round $r LSTM release variant $v: $ms"
  done
done

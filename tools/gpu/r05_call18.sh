#!/bin/bash
# round 5, call 18: persistent LSTM step counters that keep counting across calls (no memset launch per scan), 100x gradient
# fed into the loss terms: LSTM / CRNN / harness / trajectory tests, timing, launch count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c18
for i in 1 2; do timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or crnn" 2>&1 | tail -1; done
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_dp.py -q -m gpu -k "crnn or e2e_ctc or decoded or fresh_batch or harness or traj or dp_engine or full_size_step" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
timeout 200 python tools/dev/lstm_bench.py 128 2>&1 | tail -8
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['final_loss'])")
  echo "round $r: $ms"
done
rocprofv3 --kernel-trace -d gpurun_out/p_r05_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_r05_seq.log 2>&1
DB=$(find gpurun_out/p_r05_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r05_step_sequence.txt; wc -l gpurun_out/r05_step_sequence.txt
grep -c "fillBuffer\|vectorized_elementwise" gpurun_out/r05_step_sequence.txt
rm -rf gpurun_out/p_r05_seq

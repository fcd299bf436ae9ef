#!/bin/bash
# round 5, call 17: LSTM scans on weights prepared once (frozen recognizer); tiny 1x1 kernel in the STN backward: tests,
# LSTM timing, step in launch order
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r05_c17
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "lstm or crnn" > ${O}_pytest_k.log 2>&1; tail -2 ${O}_pytest_k.log
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k "crnn or e2e_ctc or decoded or fresh_batch or harness" > ${O}_pytest_m.log 2>&1; tail -2 ${O}_pytest_m.log
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2 3; do
  ms=$(timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'])")
  echo "round $r: $ms"
done
rocprofv3 --kernel-trace -d gpurun_out/p_r05_seq -o bench -- python bench.py --steps 8 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/p_r05_seq.log 2>&1
DB=$(find gpurun_out/p_r05_seq -name "*.db" | head -1)
python tools/rocpd_sequence.py $DB clip_adam 2 > gpurun_out/r05_step_sequence.txt; wc -l gpurun_out/r05_step_sequence.txt
grep -c "split_bf16\|tiny_linear" gpurun_out/r05_step_sequence.txt; grep "tiny_linear\|split_bf16" gpurun_out/r05_step_sequence.txt | cut -c1-120
rm -rf gpurun_out/p_r05_seq

#!/bin/bash
# mode-3 single-bf16 dP in the single-pass attention backward (DP1) + next-step keep bits drawn under the LSTM scan:
# ubench timing, parity legs with margins, interleaved step A/B (default | FOCR_MASK_EARLY=0 | B1_NO_DP1 library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; rm -f gpurun_out/test_margins.txt
timeout 120 build/attn_ubench 128 1 1 b1 2>&1 | grep -E "bwd1" 
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "gradients_elementwise or e2e_ctc_golden or train_mse_golden or traj_fixed" 2>&1 | tail -4
cat gpurun_out/test_margins.txt | grep -i "attention single\|mode 3\|mode=3" | head -20
B="python bench.py --steps 40 --warmup 20 --no-other-configs --no-cpu-baseline"
for r in 1 2; do
  for v in default maskearly0 nodp1; do
    case $v in
      default) env="";;
      maskearly0) env="FOCR_MASK_EARLY=0";;
      nodp1) env="FOCR_LIB=$GRAFT_REPO_ROOT/fudanocr_amd/libfocr_hip_nodp1.so";;
    esac
    ms=$(env $env timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['kernel'][:24], d['roofline']['avg_launch_ms'])")
    echo "round $r $v: $ms"
  done
done

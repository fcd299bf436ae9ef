#!/bin/bash
# round 5, call 14: final tree as the driver runs it: full `-m gpu` suite, `build(); smoke()`, default bench; golden model
# tests under the remaining A/B switches (per-layer paths after the removal of the PL kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r05_c14_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r05_c14_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r05_c14_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
for sw in "FOCR_FE_FUSED=0" "FOCR_SRB_FUSED=0" "FOCR_BN2_FUSE=0" "FOCR_WGRAD_SIDE=0"; do
  echo "== $sw: $(env $sw timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu -k 'train_mse_golden or e2e_ctc_golden or traj3' 2>&1 | tail -1)"
done
timeout 900 python bench.py > gpurun_out/r05_c14_bench.log 2>gpurun_out/r05_c14_bench.err
python - <<PY
import json
for l in open('gpurun_out/r05_c14_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(d['value'], d['ms_per_step'], r['kernel'][:30], r['frac'], r.get('executed_frac'), r['avg_launch_ms'], r.get('traffic'), d['config'].get('other_configs'), d['config'].get('mode1_ms_per_step'), (d.get('cpu_baseline') or {}).get('value'))
PY

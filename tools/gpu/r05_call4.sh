#!/bin/bash
# round 5, call 4: full GPU suite on the tree without the PL kernels / with the keep-word schedule and the 128->64 streaming
# weight gradient; smoke; profile set r05a (kernel trace, FETCH / WRITE / MFMA PMC passes, gaps) of the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
SECONDS=0
timeout 1700 python -m pytest tests -q -m gpu > gpurun_out/r05_c4_pytest.log 2>&1
echo "rc=$? wall ${SECONDS}s" >> gpurun_out/r05_c4_pytest.log; grep -E "passed|failed|^FAILED|^ERROR|^rc=" gpurun_out/r05_c4_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
bash tools/profile_round.sh r05a --no-other-configs > gpurun_out/r05a_profile.log 2>&1; tail -1 gpurun_out/r05a_kt_total.txt; head -1 gpurun_out/r05a_gaps.txt
cat gpurun_out/r05a_mfma_util.txt | head -20
python tools/rocpd_bygrid.py $(ls gpurun_out/p_r05a_kt/*.db 2>/dev/null | head -1) "" 13 2>/dev/null | head -5

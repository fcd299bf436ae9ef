#!/bin/bash
# full GPU suite + smoke + default bench on the current tree
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/test_margins.txt
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r04_c13_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r04_c13_pytest.log; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r04_c13_pytest.log | head -20
python -c "import __graft_entry__ as G; G.build(); G.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r04_c13_bench.log 2>gpurun_out/r04_c13_bench.err; tail -c 1500 gpurun_out/r04_c13_bench.log

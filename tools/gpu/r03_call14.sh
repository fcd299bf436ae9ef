#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/dev/dropout_cost.py 2>&1 | tail -2 | tee gpurun_out/c14_dropout.log

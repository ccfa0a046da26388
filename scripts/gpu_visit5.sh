#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_qsigma.py -m gpu -q --timeout 600 > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -40 $O/tests.log

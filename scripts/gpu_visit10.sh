#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/v10
mkdir -p $O
for ch in 1 2 3 4; do echo "chains $ch"; RSRL_K1_CHAINS=$ch python scripts/bench_configs.py "C2 same" | cut -c1-330; done
timeout 900 python -m pytest tests/test_gpu_parity_mc.py tests/test_gpu_parity_more.py tests/test_gpu_bitwise.py -m gpu -q --timeout 600 > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -4 $O/tests.log

#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/v11
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_golden.py -m gpu -q --timeout 300 2>&1 | tail -3

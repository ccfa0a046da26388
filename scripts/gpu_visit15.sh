#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v15
rm -rf $O; mkdir -p $O
for e in none rccl peer; do python scripts/prof_shared.py fourier $e; done
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x -k "shared or c4 or C4 or multirank or coalesc or checkpoint" > $O/pytest_shared.log 2>&1; tail -5 $O/pytest_shared.log

#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v14
rm -rf $O; mkdir -p $O
timeout 300 python scripts/bench_configs.py "C5/2" "C5'" > $O/bench_configs.jsonl 2>&1; cut -c1-250 $O/bench_configs.jsonl
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -x -k "wave or c5 or C5 or order7 or fullsize" > $O/pytest_wave.log 2>&1; tail -5 $O/pytest_wave.log

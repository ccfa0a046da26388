#!/bin/bash
# branch-free policy / TD helpers: bitwise suites + the driver's bench command + the size sweep
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v13
rm -rf $O; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-600 $O/bench_driver.json
timeout 300 python scripts/scale_n.py > $O/scale_n.jsonl 2>&1; cat $O/scale_n.jsonl
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python scripts/bench_configs.py > $O/bench_configs.jsonl 2>&1; cut -c1-200 $O/bench_configs.jsonl

#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof summary.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -m2 -E "gfx950|Marketing" > gpurun_out/device.txt 2>&1
{ nproc; python -c "import os;print(len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket"; } >> gpurun_out/device.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.log 2>&1
echo "bench exit $?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
timeout 300 python bench.py --steps-per-launch 1 --steps 3000 --warmup 300 --no-cpu-baseline > gpurun_out/bench_k1.log 2>&1
tail -2 gpurun_out/bench_k1.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/prof -name "*stats*" | head; 

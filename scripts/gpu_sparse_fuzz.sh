#!/bin/bash
# the sparse-trace agents only (FUZZ_FAMILY), after the scatter kernel's second form: short and long horizons, bitwise vs the device-order oracle; rank groups
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_sparse
FUZZ_FAMILY=sparse_lambda timeout 1500 python tests/fuzz_parity.py 1200 201 > gpurun_out/fuzz_sparse/short.log 2>&1; echo "short rc=$?"; tail -1 gpurun_out/fuzz_sparse/short.log | cut -c1-300
FUZZ_FAMILY=sparse_lambda FUZZ_LONG=1 timeout 1500 python tests/fuzz_parity.py 120 202 > gpurun_out/fuzz_sparse/long.log 2>&1; echo "long rc=$?"; tail -1 gpurun_out/fuzz_sparse/long.log | cut -c1-300
timeout 900 python tests/fuzz_ranks.py 300 203 > gpurun_out/fuzz_sparse/ranks.log 2>&1; echo "ranks rc=$?"; tail -1 gpurun_out/fuzz_sparse/ranks.log | cut -c1-300
timeout 900 python tests/fuzz_f64.py 800 204 > gpurun_out/fuzz_sparse/f64.log 2>&1; echo "f64 rc=$?"; tail -1 gpurun_out/fuzz_sparse/f64.log | cut -c1-600
for f in gpurun_out/fuzz_sparse/*.log; do grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -200 > $f.tail; mv $f.tail $f; done

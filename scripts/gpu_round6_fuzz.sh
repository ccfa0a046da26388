#!/bin/bash
# round 6's differential campaigns on the final code (sparse-trace rebuild, bf16 trace / aux agents on the wave family, the split ABI units)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_r06
for seed in 61 62 63; do timeout 1500 python tests/fuzz_parity.py 2000 $seed > gpurun_out/fuzz_r06/parity_$seed.log 2>&1; echo "parity $seed rc=$?"; tail -1 gpurun_out/fuzz_r06/parity_$seed.log | cut -c1-600; done
timeout 900 python tests/fuzz_f64.py 2000 64 > gpurun_out/fuzz_r06/f64.log 2>&1; echo "f64 rc=$?"; tail -1 gpurun_out/fuzz_r06/f64.log | cut -c1-600
timeout 900 python tests/fuzz_abi.py 2000 65 > gpurun_out/fuzz_r06/abi.log 2>&1; echo "abi rc=$?"; tail -1 gpurun_out/fuzz_r06/abi.log | cut -c1-600
timeout 900 python tests/fuzz_ranks.py 400 66 > gpurun_out/fuzz_r06/ranks.log 2>&1; echo "ranks rc=$?"; tail -1 gpurun_out/fuzz_r06/ranks.log | cut -c1-600
for f in gpurun_out/fuzz_r06/*.log; do grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -400 > $f.tail; mv $f.tail $f; done

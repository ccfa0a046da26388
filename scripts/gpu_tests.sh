#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 600 "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -60

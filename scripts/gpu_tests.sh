#!/bin/bash
set -u
mkdir -p gpurun_out
# (--timeout needs the pytest-timeout plugin: passed only where it is installed)
TO=$(python -c "import pytest_timeout" 2>/dev/null && echo "--timeout 600")
timeout 3000 python -m pytest tests -m gpu -q -x --durations=12 $TO "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" gpurun_out/pytest_gpu.log | tail -60

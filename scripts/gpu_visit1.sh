#!/bin/bash
# round 2, visit 1: the new tests first, then the whole GPU suite, then the bench under the driver's invocation
set -u
mkdir -p gpurun_out/v1
timeout 900 python -m pytest tests/test_gpu_bitwise.py tests/test_gpu_multirank.py -m gpu -q -x --timeout 600 > gpurun_out/v1/new_tests.log 2>&1
echo "exit $?" >> gpurun_out/v1/new_tests.log
tail -30 gpurun_out/v1/new_tests.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/v1/pytest_gpu.log 2>&1
echo "exit $?" >> gpurun_out/v1/pytest_gpu.log
tail -15 gpurun_out/v1/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/v1/bench_driver.json 2> gpurun_out/v1/bench_driver.err
tail -c 3000 gpurun_out/v1/bench_driver.json; tail -5 gpurun_out/v1/bench_driver.err

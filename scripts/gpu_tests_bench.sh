#!/bin/bash
set -u
bash scripts/gpu_tests.sh
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench.log; cat gpurun_out/bench.log | cut -c1-4000

#!/bin/bash
set -u
bash scripts/gpu_tests.sh
python scripts/bench_configs.py "C2" 2>&1 | cut -c1-330

#!/bin/bash
for i in 1 2 3; do timeout 300 python scripts/bench_configs.py "L1" 2>&1 | grep config | cut -c100-200; done
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "lambda or agents" 2>&1 | grep -a "passed\|failed"

#!/bin/bash
python scripts/prof_shared.py tile none
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/full.log 2>&1; echo "exit $?" >> gpurun_out/full.log; grep -E "passed|failed|exit" gpurun_out/full.log

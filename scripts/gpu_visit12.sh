#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_bitwise.py -m gpu -q --timeout 600 -k "c3_shared or c4_shared" 2>&1 | tail -15
python scripts/prof_shared.py tile none
timeout 900 python -m pytest tests/test_gpu_parity_more.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_parity_pal.py -m gpu -q --timeout 600 2>&1 | tail -4

#!/bin/bash
python scripts/prof_shared.py tile none
timeout 600 python -m pytest tests/test_gpu_bitwise.py tests/test_gpu_parity_more.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "tile or c3 or C3 or shared" 2>&1 | tail -3

#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_bitwise.py -m gpu -q --timeout 600 -k "c4_shared" 2>&1 | tail -15

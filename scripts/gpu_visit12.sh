#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "sampled" 2>&1 | tail -5

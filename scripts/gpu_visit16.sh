#!/bin/bash
python scripts/prof_shared.py tile none
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "tile or c3 or C3" 2>&1 | tail -3

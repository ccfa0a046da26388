#!/bin/bash
for i in 1 2; do python scripts/prof_shared.py tile none; RSRL_TILE_FUSED_SCATTER=1 python scripts/prof_shared.py tile none | sed 's/^/fused-scatter /'; done
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x -k "tile or c3 or C3 or too_large or handle" 2>&1 | grep -a "passed\|failed\|^FAILED"

#!/bin/bash
set -u
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "tile or c3 or C3 or too_large or handle or shared" 2>&1 | grep -a "passed\|failed\|^FAILED\|Error" | tail
for i in 1 2 3; do python scripts/prof_shared.py tile none; RSRL_TILE_SEPARATE_APPLY=1 python scripts/prof_shared.py tile none | sed 's/^/separate-apply /'; done 2>&1 | grep us/step

#!/bin/bash
# round 3, visit 3: persistent shared-W kernel -- bitwise tests, then timing against the one-launch-per-step path
set -u
mkdir -p gpurun_out/r3v3
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 -k "c4 or shared or multirank or size1 or g_ranks or missing_peer or two_processes" 2>&1 | tail -15
for i in 1 2; do
  timeout 120 python scripts/prof_shared.py fourier none
  RSRL_NO_PERSIST=1 timeout 120 python scripts/prof_shared.py fourier none | sed 's/^/no-persist /'
  timeout 120 python scripts/prof_shared.py fourier peer
  RSRL_NO_PERSIST=1 timeout 120 python scripts/prof_shared.py fourier peer | sed 's/^/no-persist /'
done

#!/bin/bash
# where the persistent shared-W step goes: builds with pieces compiled out (RSRL_PERSIST_ABLATE bit 0 = no exchange, 1 = no MFMA
# chain, 2 = no learner work); the variants were built on the CPU box by scripts/build_persist_variants.py
set -u
for v in 0 1 2 4 3 5 6 7; do
  lib=rsrl_amd/lib/librsrl_hip_ab$v.so
  [ -f $lib ] || continue
  for i in 1 2; do RSRL_HIP_LIB=$PWD/$lib timeout 120 python scripts/prof_shared.py fourier none | sed "s/^/ablate=$v /"; done
done

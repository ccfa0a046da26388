#!/bin/bash
# A/B of k_shared_step variants (libraries under rsrl_amd/lib/variants), interleaved rounds
for round in 1 2 3; do
for v in base $(ls $GRAFT_REPO_ROOT/rsrl_amd/lib/variants | sed 's/.so//'); do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python scripts/prof_shared.py fourier none | sed "s/^/$v /"
done
done

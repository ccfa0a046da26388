#!/bin/bash
# timing experiment: k_shared_step with pieces compiled out (variant libraries under rsrl_amd/lib/variants)
for v in base $(ls $GRAFT_REPO_ROOT/rsrl_amd/lib/variants | sed 's/.so//'); do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python scripts/prof_shared.py fourier none | sed "s/^/$v /"
done

#!/bin/bash
# timing experiment: shared tile-coding step with pieces compiled out (variant libraries under rsrl_amd/lib/variants)
for v in base nogatomic; do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python scripts/prof_shared.py tile none | sed "s/^/$v /"
done

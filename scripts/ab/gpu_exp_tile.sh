#!/bin/bash
# timing experiment: shared tile-coding step (C3) with pieces compiled out (variant libraries under rsrl_amd/lib/variants,
# built by scripts/build_variants.py), plus the per-kernel averages of the base build
for round in 1 2; do
for v in base $(ls $GRAFT_REPO_ROOT/rsrl_amd/lib/variants 2>/dev/null | sed 's/.so//'); do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python scripts/prof_shared.py tile none | sed "s/^/$v /"
done
done

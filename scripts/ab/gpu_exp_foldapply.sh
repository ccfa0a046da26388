#!/bin/bash
# timing experiment (results are NOT valid weights): C3 with the apply folded into the step kernel's gathers, emulated -- one
# 64-bit table entry gathered and converted per gathered weight (variant foldtab), the third launch skipped (RSRL_TILE_SKIP_APPLY),
# one or four copies of the scatter's table (RSRL_TILE_REPLICAS)
cd "$(dirname "$0")/.."
for round in 1 2; do
  python scripts/prof_shared.py tile none | sed "s/^/base            /"
  RSRL_TILE_SKIP_APPLY=1 python scripts/prof_shared.py tile none | sed "s/^/skip-apply      /"
  RSRL_HIP_LIB=rsrl_amd/lib/variants/foldtab.so RSRL_TILE_SKIP_APPLY=1 python scripts/prof_shared.py tile none | sed "s/^/fold R=4        /"
  RSRL_HIP_LIB=rsrl_amd/lib/variants/foldtab.so RSRL_TILE_SKIP_APPLY=1 RSRL_TILE_REPLICAS=1 python scripts/prof_shared.py tile none | sed "s/^/fold R=1        /"
  RSRL_HIP_LIB=rsrl_amd/lib/variants/foldtab.so RSRL_TILE_SKIP_APPLY=1 RSRL_TILE_REPLICAS=2 python scripts/prof_shared.py tile none | sed "s/^/fold R=2        /"
done 2>&1 | grep -v "^.*RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tee gpurun_out/exp_foldapply.txt

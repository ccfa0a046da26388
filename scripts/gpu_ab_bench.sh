#!/bin/bash
# interleaved A/B of the headline bench (short regions) between the product library and the variants under rsrl_amd/lib/variants
set -u
for round in 1 2 3; do
for v in base $(ls $GRAFT_REPO_ROOT/rsrl_amd/lib/variants 2>/dev/null | sed 's/.so//'); do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python bench.py --steps 1024 --warmup 64 --regions 3 --region-seconds 0.4 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg --no-nocoalesce-leg 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '%.4g'%d['value'], 'launch_ms %.4f'%d['roofline']['avg_launch_ms'])"
done
done

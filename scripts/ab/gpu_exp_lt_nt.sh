#!/bin/bash
# round 3: non-temporal accesses in the table sweep of k_lambda_tile; variants built by scripts/build_variants.py (lt0: -DRSRL_LT_NT=0,
# d3: -DRSRL_LT_NT=3) plus a byte copy of the product library -- is the difference the code or the run?
cd "$(dirname "$0")/.."
for r in 1 2 3; do for v in base copy d3 lt0; do
  lib=rsrl_amd/lib/librsrl_hip.so; [ $v != base ] && lib=rsrl_amd/lib/variants/$v.so
  RSRL_HIP_LIB=$lib python scripts/bench_configs.py L2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['kernel'], '%.4g' % d['env_steps_per_s'], '%.3f' % d['frac_of_8TBps'])"
done; done

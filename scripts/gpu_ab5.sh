#!/bin/bash
set -u
for round in 1 2 3; do
for v in base cheaprng; do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so; fi
  python scripts/bench_configs.py "fused 256" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'launch us %.2f  steps/s %.4g'%(d.get('avg_launch_us',-1), d.get('env_steps_per_s',-1)), d.get('error',''))
"
done
done

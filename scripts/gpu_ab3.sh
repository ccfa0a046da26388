#!/bin/bash
set -u
for round in 1 2; do
for v in r0s0 r0s1 r1s0 r1s1; do
  export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so
  python scripts/bench_configs.py "C2 same" 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', 'launch us %.2f  steps/s %.3g'%(d.get('avg_launch_us',-1), d.get('env_steps_per_s',-1)), d.get('error',''))
"
done
done

#!/bin/bash
set -u
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 value %.4g ms/launch %.5f frac %.3f rollout %.1f'%(d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['greedy_rollout_mean_n_states']))
    elif 'rror' in l: print(l.strip())
"; }
for round in 1 2 3; do
  for v in base storeall; do
    if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so; fi
    python bench.py --no-cpu-baseline --steps-per-launch 1 --steps 5000 --warmup 500 2>&1 | summ "$v-k1 r$round"
  done
done

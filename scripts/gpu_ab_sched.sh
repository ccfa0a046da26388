#!/bin/bash
# A/B of build variants of the fused kernel (interleaved rounds, one box visit) + bitwise identity against the base build
set -u
mkdir -p gpurun_out
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 value %.4g ms/launch %.5f'%(d['value'],d['roofline']['avg_launch_ms']))
    elif 'rror' in l: print(l.strip())
"; }
VARIANTS=${VARIANTS:-base maxmem pk pkmem pkilp}
for round in 1 2 3; do
  for v in $VARIANTS; do
    if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so; fi
    python bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 10240 --warmup 1024 2>&1 | summ "$v-fused r$round"
  done
done
for v in $VARIANTS; do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so; fi
  python - <<PY
import rsrl_amd as ra, hashlib, numpy as np
out = []
for kw in (dict(order=5, algo=0, policy=1), dict(order=3, algo=1, policy=1), dict(order=5, algo=2, policy=2, alpha=0.7), dict(domain=1, order=1, algo=5, policy=1, alpha=0.5)):
    with ra.Context(n_envs=4096, epsilon=0.1, seed=5, max_episode_steps=200, **kw) as c:
        c.reset(); c.train(600)
        out.append(hashlib.md5(c.get_weights(77).tobytes() + c.states.tobytes() + c.actions.tobytes()).hexdigest()[:10])
print("$v digests", out)
PY
done

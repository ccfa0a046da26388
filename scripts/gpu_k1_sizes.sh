#!/bin/bash
# the two learner-major single-step kernels (RSRL_K1_QUAD = 0: one lane per learner, 1: four) at several sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for q in 0 1; do
  for n in 65536 131072 262144 524288 1048576; do
    RSRL_K1_QUAD=$q python - <<PY
import json, time, rsrl_amd as ra
n=$n
c = ra.Context(n_envs=n, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1)
c.reset(); c.train(300, want_stats=False); c.sync()
c.timing_enable(True)
steps = 2000 if n <= 262144 else 500
t0=time.perf_counter(); c.train(steps, want_stats=False); c.sync(); dt=time.perf_counter()-t0
ms, cnt, kn = c.timing_read()
us = ms*1e3/max(1,cnt)
print(json.dumps({"lanes_per_learner": 4 if $q else 1, "n": n, "kernel": kn, "us_per_step_wall": round(dt/steps*1e6,2), "avg_launch_us": round(us,2),
  "frac_8TBps_kernel": round(608*n/us/8e6,3)}))
c.close()
PY
  done
done 2>&1 | tee gpurun_out/k1_sizes.txt

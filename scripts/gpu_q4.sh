#!/bin/bash
# round 3: the four-lanes-per-learner streaming kernel against the one-lane kernel -- tests, then rates at several sizes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity_mc.py tests/test_gpu_parity_more.py -x -q -m gpu 2>&1 | tail -5
for q in 1 0; do
  for n in 65536 262144 1048576; do
    RSRL_K1_QUAD=$q python - <<PY
import json, time, rsrl_amd as ra
n=$n
c = ra.Context(n_envs=n, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1)
c.reset(); c.train(300, want_stats=False); c.sync()
c.timing_enable(True)
steps = 2000 if n <= 262144 else 500
t0=time.perf_counter(); c.train(steps, want_stats=False); c.sync(); dt=time.perf_counter()-t0
ms, cnt, kn = c.timing_read()
print(json.dumps({"quad": $q, "n": n, "kernel": kn, "us_per_step_wall": dt/steps*1e6, "avg_launch_us": ms*1e3/max(1,cnt), "launches": cnt,
  "frac_8TBps_wall": 608*n*steps/dt/8e12, "env_steps_per_s": n*steps/dt}))
c.close()
PY
  done
done 2>&1 | tee gpurun_out/q4_rates.txt

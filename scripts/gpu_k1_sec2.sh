#!/bin/bash
# round 3: k_step_reg_lm with cooperative whole-sector stores (variant dma) against the default
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RSRL_HIP_LIB=rsrl_amd/lib/variants/dma.so RSRL_K1_QUAD=0 python -m pytest tests/test_gpu_parity_mc.py -x -q -m gpu -k "single_step" 2>&1 | tail -3
for v in dma base; do
  for n in 65536 131072 262144 1048576; do
    lib=rsrl_amd/lib/variants/$v.so
    [ $v = base ] && lib=rsrl_amd/lib/librsrl_hip.so
    RSRL_HIP_LIB=$lib RSRL_K1_QUAD=0 python - <<PY
import json, time, rsrl_amd as ra
n=$n
c = ra.Context(n_envs=n, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1)
c.reset(); c.train(300, want_stats=False); c.sync()
c.timing_enable(True)
steps = 2000 if n <= 262144 else 500
t0=time.perf_counter(); c.train(steps, want_stats=False); c.sync(); dt=time.perf_counter()-t0
ms, cnt, kn = c.timing_read()
print(json.dumps({"variant": "$v", "n": n, "kernel": kn, "us_per_step_wall": round(dt/steps*1e6,2), "avg_launch_us": round(ms*1e3/max(1,cnt),2),
  "frac_8TBps_wall": round(608*n*steps/dt/8e12,3)}))
c.close()
PY
  done
done 2>&1 | tee gpurun_out/k1_dma.txt

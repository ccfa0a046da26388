#!/bin/bash
# fuse depth of the register-family loop under the driver's invocation (20-step calls, coalesced)
for round in 1 2; do
for d in 1024 2048 4096 8192; do
  RSRL_FUSE_DEPTH=$d python bench.py --steps 20 --warmup 5 --regions 3 --region-seconds 0.5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg --no-nocoalesce-leg 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('depth $d', '%.4g'%d['value'], 'launch_ms %.4f'%d['roofline']['avg_launch_ms'], 'spl %.0f'%d['config']['steps_per_launch'])"
done
done

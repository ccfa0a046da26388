#!/bin/bash
# round 3: wave family with the learner index made provably wave-uniform (variant wuni) against the committed build
cd "$(dirname "$0")/.."
RSRL_HIP_LIB=rsrl_amd/lib/variants/wuni.so python -m pytest tests/test_gpu_bitwise.py tests/test_gpu_parity_wave.py tests/test_gpu_wave_lambda.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
  for v in base wuni; do
    lib=rsrl_amd/lib/librsrl_hip.so; [ $v = wuni ] && lib=rsrl_amd/lib/variants/wuni.so
    RSRL_HIP_LIB=$lib python scripts/bench_configs.py C5 L3 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$v', d['config'][:40], '%.4g' % d['env_steps_per_s'], d['kernel'])"
  done
done 2>&1 | tee gpurun_out/exp_wave_uniform.txt

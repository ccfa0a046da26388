#!/bin/bash
# A/B of kernel variants (interleaved rounds in one box visit) + parity tests + K=1 profile
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
summ() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 value %.4g ms/launch %.5f frac %.3f rollout %.1f'%(d['value'],d['roofline']['avg_launch_ms'],d['roofline']['frac'],d['greedy_rollout_mean_n_states']))
    elif 'rror' in l: print(l.strip())
"; }
for round in 1 2 3; do
  for v in base rank1; do
    if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$PWD/rsrl_amd/lib/variants/$v.so; fi
    python bench.py --no-cpu-baseline --steps 10000 --warmup 1000 2>&1 | summ "$v-fused r$round"
    python bench.py --no-cpu-baseline --steps-per-launch 1 --steps 5000 --warmup 500 2>&1 | summ "$v-k1 r$round"
  done
done
unset RSRL_HIP_LIB
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_k1 -o k1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps-per-launch 1 --steps 3000 --warmup 300 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_k1.log 2>&1

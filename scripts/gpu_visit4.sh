#!/bin/bash
# round 2, visit 4: the single-launch dense shared-W step
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v4
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_parity_more.py tests/test_gpu_fullsize.py tests/test_gpu_parity_pal.py -m gpu -q --timeout 600 > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -30 $O/tests.log
for e in none rccl peer; do python scripts/prof_shared.py fourier $e; done
cd /tmp
for e in none peer; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fourier_$e -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier $e > $O/prof_fourier_$e.log 2>&1
  head -6 $O/prof_fourier_$e/s_kernel_stats.csv | cut -c1-220
done

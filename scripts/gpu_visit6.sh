#!/bin/bash
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v6
mkdir -p $O
./scripts/ubench/lds_atomic > $O/lds_atomic.txt
python scripts/prof_shared.py tile none
timeout 900 python -m pytest tests/test_gpu_parity_more.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -q --timeout 600 > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -5 $O/tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tile_none -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py tile none > $O/prof_tile.log 2>&1
head -4 $O/prof_tile_none/s_kernel_stats.csv | cut -c1-60,180-300

#!/bin/bash
# round 2, visit 2: cross-process experiment, fixed multi-rank tests, bitwise tests on the rebuilt fused kernel, bench lines, shared-W profiles
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v2
mkdir -p $O
timeout 200 python scripts/ubench/xproc_spin.py 1.0 > $O/xproc.log 2>&1; tail -30 $O/xproc.log
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_bitwise.py -m gpu -q --timeout 400 > $O/tests.log 2>&1; echo "exit $?" >> $O/tests.log; tail -25 $O/tests.log
timeout 300 python -m pytest tests/test_gpu_parity_mc.py tests/test_gpu_parity_more.py -m gpu -q -x --timeout 400 > $O/tests2.log 2>&1; echo "exit $?" >> $O/tests2.log; tail -8 $O/tests2.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-shared-leg --no-streaming-leg > $O/bench_k20.json 2> $O/bench_k20.err; cut -c1-700 $O/bench_k20.json
timeout 300 python bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg > $O/bench_default.json 2> $O/bench_default.err; cut -c1-700 $O/bench_default.json
for n in 131072 262144 1048576; do timeout 300 python bench.py --envs $n --no-cpu-baseline --no-shared-leg --no-streaming-leg > $O/bench_n$n.json 2> $O/bench_n$n.err; cut -c1-400 $O/bench_n$n.json; done
cd /tmp
for e in none rccl peer; do
  python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier $e
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fourier_$e -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier $e > $O/prof_fourier_$e.log 2>&1
done
python $GRAFT_REPO_ROOT/scripts/prof_shared.py tile none
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tile_none -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py tile none > $O/prof_tile_none.log 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
cd $O; find . -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -8 $f | cut -c1-260; done
du -sh $O

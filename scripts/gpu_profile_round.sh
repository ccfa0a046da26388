#!/bin/bash
# Round profile: rocprofv3 --kernel-trace --stats of the bench command + PMC passes for the two MountainCar kernels.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/stats.log
K1="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps-per-launch 1 --steps 400 --warmup 100"
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 2560 --warmup 256"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/k1_$i -o p -- $K1 > $OUT/k1_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/fu_$i -o p -- $FU > $OUT/fu_$i.log 2>&1
done
ls $OUT $OUT/stats | head -40

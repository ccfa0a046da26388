#!/bin/bash
# round 3: PMC passes (each counter group in its own rocprofv3 run, --kernel-trace only) of the fused kernel at 256 steps per launch
# and of the single-step streaming kernel -> gpurun_out/pmc_r03/{fu_*,k1_*}; scripts/summarize_r03.py turns them into
# profiles/r03_pmc_raw.json, profiles/isa_mix.json and profiles/pmc_traffic.json
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_r03
[ "${1:-}" = "k1only" ] || rm -rf $O
mkdir -p $O
cd /tmp
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --no-config-legs --no-nocoalesce-leg --regions 1 --region-seconds 0.2 --steps 2560 --warmup 256 --steps-per-launch 256"
# the streaming kernel: plain launches (RSRL_NO_GRAPH=1; graph replays under counter collection crashed the profiler's host side in this round)
cat > /tmp/k1_pmc.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import rsrl_amd as ra
c = ra.Context(n_envs=65536, policy=1, epsilon=0.1, max_episode_steps=1000, steps_per_launch=1)
c.reset(); c.train(100, want_stats=False); c.sync(); c.train(300, want_stats=False); c.sync(); c.close()
PY
K1="env RSRL_NO_GRAPH=1 python /tmp/k1_pmc.py"
if [ "${1:-}" = "k1only" ]; then SKIP_FU=1; rm -rf $O/k1_*; else SKIP_FU=0; fi
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  [ $SKIP_FU = 1 ] || timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/fu_$i -o p -- $FU > $O/fu_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/k1_$i -o p -- $K1 > $O/k1_$i.log 2>&1
done
find $O -name "*_agent_info.csv" -delete
# rocprofv3 nests its output (<dir>/<host>/<pid>_...csv): flatten so that the summariser finds <dir>/*counter_collection.csv
for d in $O/fu_* $O/k1_*; do [ -d $d ] && find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \; ; done
ls $O/fu_1 $O/k1_1 | head; du -sh $O

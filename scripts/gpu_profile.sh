#!/bin/bash
# The round's evidence, one visit:  gpurun -- 'bash scripts/gpu_profile.sh [legs...]'   (default: every leg of scripts/profile_leg.py)
#   gpurun_out/prof/bench_driver.json          the driver's invocation, plain
#   gpurun_out/prof/kt_driver/                 the same command under rocprofv3 --kernel-trace --stats (legs off: one population)
#   gpurun_out/prof/kt_legs/                   ... with every secondary leg on
#   gpurun_out/prof/<leg>/{kt,p1..p5}/         per leg: kernel trace + one rocprofv3 run PER counter group (--kernel-trace only, as the
#                                              guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass)
# scripts/summarize_profile.py <tag> turns it into profiles/<tag>_* + profiles/isa_mix.json + profiles/pmc_traffic.json (what bench.py reads).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
[ "${PROFILE_LEGS_ONLY:-0}" = "1" ] || rm -rf $O
mkdir -p $O
LEGS=${*:-fused stream stream1m persist perstep tile wave}
cd /tmp
if [ "${PROFILE_LEGS_ONLY:-0}" = "1" ]; then SKIP_BENCH=1; else SKIP_BENCH=0; fi
[ $SKIP_BENCH = 1 ] || {
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_driver -o k -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg > $O/bench_driver_profiled.json 2> $O/kt_driver.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1024 -o k -- python $R/bench.py --gpus 1 --steps 1024 --warmup 5 --no-cpu-baseline --no-config-legs --no-shared-leg --no-streaming-leg --no-nocoalesce-leg > $O/bench_1024_profiled.json 2> $O/kt_1024.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_legs -o k -- python $R/bench.py --gpus 1 --steps 1024 --warmup 5 --no-cpu-baseline --regions 1 --region-seconds 0.2 --no-nocoalesce-leg > $O/bench_legs_profiled.json 2> $O/kt_legs.log
}
G1="FETCH_SIZE"
G2="WRITE_SIZE"
G3="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
G4="SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
G5="SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FLOPS_FP64"
for leg in $LEGS; do
  rm -rf $O/$leg; mkdir -p $O/$leg
  # plain launches under counter collection (graph replays crashed the profiler's host side in round 3)
  CMD="env RSRL_NO_GRAPH=1 python $R/scripts/profile_leg.py $leg"
  python $R/scripts/profile_leg.py $leg > $O/$leg/plain.json 2> $O/$leg/plain.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$leg/kt -o k -- $CMD > $O/$leg/kt.json 2> $O/$leg/kt.log
  i=0
  for grp in "$G1" "$G2" "$G3" "$G4" "$G5"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/$leg/p$i -o p -- $CMD > $O/$leg/p$i.json 2> $O/$leg/p$i.log
  done
done
# what the profiled processes loaded: the sha256 of every profiled kernel's machine code (bench.py checks the committed constants against it)
(cd $R && python -c "
import json
from rsrl_amd import _kdigest, _build
print(json.dumps(_kdigest.kernel_digests(_build.LIB_PATH, ['k_train_reg', 'k_step_reg_lm', 'k_step_reg_q4', 'k_shared_persist', 'k_shared_step', 'k_shared_ca', 'k_tile_scatter', 'k_apply_rep', 'k_train_wave', 'k_train_wave_pk'])))" > $O/kernel_digests.json)
find $O -name "*_agent_info.csv" -delete
# rocprofv3 nests its output (<dir>/<host>/<pid>_...csv): flatten
for d in $(find $O -mindepth 1 -maxdepth 2 -type d -name "kt*" -o -mindepth 1 -maxdepth 2 -type d -name "p[0-9]"); do find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \; 2>/dev/null; done
du -sh $O; for leg in $LEGS; do cat $O/$leg/plain.json; done

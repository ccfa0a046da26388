#!/bin/bash
# PMC passes (each counter group in its own rocprofv3 run, --kernel-trace only) for the K=1 and fused kernels
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp
K1="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps-per-launch 1 --steps 400 --warmup 100"
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2560 --warmup 256"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/k1_$i -o p -- $K1 > $OUT/k1_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/fu_$i -o p -- $FU > $OUT/fu_$i.log 2>&1
done
cd $OUT; find . -name "*.csv" | head -40; du -sh .

#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1b
mkdir -p $OUT
cd /tmp
K1="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --steps-per-launch 1 --steps 400 --warmup 100"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/k1_$i -o p -- $K1 > $OUT/k1_$i.log 2>&1
done
python - <<PY
import csv, collections
for i in range(1,5):
    agg=collections.defaultdict(list); dur=[]
    for r in csv.DictReader(open("$OUT/k1_%d/p_counter_collection.csv"%i)):
        if 'k_step_reg' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value'])); dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    for k,v in agg.items():
        v=v[len(v)//4:]; print("%-24s mean=%.6g"%(k,sum(v)/len(v)))
    print("  kernel dur us", sum(dur)/max(1,len(dur)))
PY

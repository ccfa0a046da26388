#!/bin/bash
# PMC passes of the fused kernel only (instruction mix, wait/active cycles)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmcf
rm -rf $O; mkdir -p $O
cd /tmp
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 2560 --warmup 256 --repeats 4 --steps-per-launch 256"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/fu_$i -o p -- $FU > $O/fu_$i.log 2>&1
done
find $O -name "*_agent_info.csv" -delete
python - <<'PY'
import csv, glob, os, collections
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmcf"
for f in sorted(glob.glob(O+"/fu_*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_train_reg" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(os.path.basename(os.path.dirname(os.path.dirname(f))) if False else "", k, sum(v)/len(v), len(v))
PY

#!/bin/bash
# round 2, visit 3: whole GPU suite, bench lines (driver invocation + default), PMC passes of the fused kernel (instruction classes, traffic)
set -u
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/v3
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; tail -12 $O/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cut -c1-900 $O/bench_driver.json; tail -3 $O/bench_driver.err
RSRL_NO_COALESCE=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-shared-leg --no-streaming-leg > $O/bench_k20_nocoalesce.json 2>&1; cut -c1-500 $O/bench_k20_nocoalesce.json
cd /tmp
FU="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 2560 --warmup 256 --repeats 4"
K20="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-shared-leg --no-streaming-leg --steps 20 --warmup 5 --repeats 200"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "SQ_WAVES SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/fu_$i -o p -- $FU > $O/fu_$i.log 2>&1
  if [ $i -ge 3 ]; then RSRL_NO_COALESCE=1 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/k20_$i -o p -- $K20 > $O/k20_$i.log 2>&1; fi
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_driver -o b -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_prof.json 2> $O/stats_driver.log
head -12 $O/stats_driver/b_kernel_stats.csv | cut -c1-200
python3 - <<PY
import csv, glob, collections
O="$O"
for d in sorted(glob.glob(O+"/fu_*")+glob.glob(O+"/k20_*")):
    f=glob.glob(d+"/*counter_collection.csv")
    if not f: print(d,"no csv"); continue
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        if "k_train_reg" not in r["Kernel_Name"]: continue
        per[r["Counter_Name"]][int(r["Dispatch_Id"])]+=float(r["Counter_Value"])
    out={}
    for n,by in per.items():
        ids=sorted(by)[2:] or sorted(by)
        out[n]=sum(by[i] for i in ids)/len(ids)
    print(d.split("/")[-1], {k: round(v,1) for k,v in out.items()})
PY
du -sh $O

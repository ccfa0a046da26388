#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python scripts/prof_shared.py fourier; python scripts/prof_shared.py tile
cd /tmp
for m in fourier tile; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_shared_$m -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py $m > /dev/null 2>&1
python - <<PY
import csv, collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_shared_$m/s_kernel_trace.csv")):
    agg[r['Kernel_Name'][:90]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])): print("%-92s n=%5d avg %8.2f us total %9.1f"%(k,len(v),sum(v)/len(v),sum(v)))
PY
done

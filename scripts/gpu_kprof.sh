#!/bin/bash
# per-kernel profile of one command on the GPU box:  gpu_kprof.sh <tag> [--pmc "C1 C2 ..."]... -- <command ...>
#   gpurun_out/kprof/<tag>/stats.txt   rocprofv3 --kernel-trace: calls, average / total duration per kernel
#   gpurun_out/kprof/<tag>/pmc.txt     per kernel and PMC pass: mean of every counter's per-dispatch sum, plus registers / LDS
set -u
export TMPDIR=/tmp
tag=$1; shift
passes=()
while [ "$1" != "--" ]; do if [ "$1" = "--pmc" ]; then passes+=("$2"); shift 2; else shift; fi; done
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/kprof/$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o k -- "$@" > $OUT/kt.log 2>&1
python3 - "$OUT" <<'PY' | tee $OUT/stats.txt
import csv, collections, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(list); meta = {}
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]
    agg[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    meta[k] = (r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"), r.get("SGPR_Count", "?"), r.get("LDS_Block_Size", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-100s n=%6d avg %9.2f us  total %9.2f ms %5.1f%%  vgpr %s agpr %s sgpr %s lds %s grid %s wg %s" % ((k, len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot) + meta[k]))
PY
i=0
for grp in "${passes[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p -- "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" <<'PY' | tee $OUT/pmc.txt
import csv, collections, glob, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        per[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for k, cs in per.items():
        line = []
        for c, by in sorted(cs.items()):
            ids = sorted(by)[2:] or sorted(by)
            line.append("%s=%.4g" % (c, sum(by[j] for j in ids) / len(ids)))
        print("%-60s %s" % (k, "  ".join(line)))
PY

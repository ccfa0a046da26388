#!/bin/bash
# the trait-loop part of scripts/gpu_round6_profile.sh alone (fused / unfused / generic: kernel trace + FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/profiles_r06/r06_kernel_stats_trait.md
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof/trait
cd /tmp
for variant in fused unfused generic; do
  rm -rf $O/$variant; mkdir -p $O/$variant
  if [ $variant = unfused ]; then export RSRL_NO_TRAIT_DEFER=1; else unset RSRL_NO_TRAIT_DEFER; fi
  SPL=1; [ $variant = generic ] && SPL=0
  CMD="python $R/scripts/trait_loop.py 65536 300 $SPL"
  $CMD 2> $O/$variant/plain.err | grep "^{" > $O/$variant/plain.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$variant/kt -o k -- $CMD > $O/$variant/kt.json 2> $O/$variant/kt.log
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$variant/p1 -o p -- $CMD > $O/$variant/p1.json 2> $O/$variant/p1.log
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$variant/p2 -o p -- $CMD > $O/$variant/p2.json 2> $O/$variant/p2.log
done
unset RSRL_NO_TRAIT_DEFER
for d in $(find $O -mindepth 1 -maxdepth 3 -type d \( -name "kt" -o -name "p[0-9]" \)); do find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \; 2>/dev/null; done
find $O -name "*_agent_info.csv" -delete
cd $R
python scripts/summarize_trait.py r06 > gpurun_out/summarize_trait_r06.log 2>&1
mkdir -p gpurun_out/profiles_r06; cp profiles/r06_kernel_stats_trait.md gpurun_out/profiles_r06/
find gpurun_out/prof -name "*kernel_trace.csv" -delete
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
cat profiles/r06_kernel_stats_trait.md

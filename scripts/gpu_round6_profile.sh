#!/bin/bash
# Round 6's evidence in ONE visit:  gpurun -- 'bash scripts/gpu_round6_profile.sh'
#   1. scripts/gpu_profile.sh                  the bench command plain and under rocprofv3, every leg's kernel trace + one run per counter group
#   2. the trait-granular loop (scripts/trait_loop.py) fused / unfused: kernel trace + FETCH_SIZE / WRITE_SIZE passes      -> profiles/r06_kernel_stats_trait.md
#   3. the widened rows' kernels (scripts/profile_round5_kernels.py) timed bare and under the kernel trace                  -> profiles/r06_new_kernels*
#   4. scripts/summarize_profile.py r06 ON THE BOX (isa_mix.json / pmc_traffic.json stamped with the loaded library's kernel digests), the support matrix,
#      then the driver's bench command once more: the line whose roofline constants match the binary                        -> profiles/r06_bench_driver.json
# Everything to be tracked is copied to gpurun_out/profiles_r06/ (only gpurun_out/ travels back).
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
bash scripts/gpu_profile.sh > gpurun_out/gpu_profile.log 2>&1
O=$R/gpurun_out/prof/trait
cd /tmp
for variant in fused unfused generic; do
  mkdir -p $O/$variant
  if [ $variant = unfused ]; then export RSRL_NO_TRAIT_DEFER=1; else unset RSRL_NO_TRAIT_DEFER; fi
  SPL=1; [ $variant = generic ] && SPL=0      # steps_per_launch = 1: the learner-major layout the fast trait kernels (k_trait_lm) read
  CMD="python $R/scripts/trait_loop.py 65536 300 $SPL"
  $CMD 2> $O/$variant/plain.err | grep "^{" > $O/$variant/plain.json
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$variant/kt -o k -- $CMD > $O/$variant/kt.json 2> $O/$variant/kt.log
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$variant/p1 -o p -- $CMD > $O/$variant/p1.json 2> $O/$variant/p1.log
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/$variant/p2 -o p -- $CMD > $O/$variant/p2.json 2> $O/$variant/p2.log
done
unset RSRL_NO_TRAIT_DEFER
NK=$R/gpurun_out/prof/newk
mkdir -p $NK
python $R/scripts/profile_round5_kernels.py 2> /dev/null | grep "^{" > $R/gpurun_out/r06_new_kernels.jsonl
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $NK/kt -o k -- python $R/scripts/profile_round5_kernels.py > $NK/kt.json 2> $NK/kt.log
for d in $(find $O $NK -mindepth 1 -maxdepth 3 -type d \( -name "kt" -o -name "p[0-9]" \)); do find $d -mindepth 2 -name "*.csv" -exec mv {} $d/ \; 2>/dev/null; done
find $R/gpurun_out/prof -name "*_agent_info.csv" -delete
cd $R
python scripts/summarize_profile.py r06 > gpurun_out/summarize_r06.log 2>&1
python scripts/summarize_trait.py r06 > gpurun_out/summarize_trait_r06.log 2>&1
python scripts/support_matrix.py > profiles/r06_support_matrix.md 2> /dev/null
cp $NK/kt/*kernel_stats.csv profiles/r06_new_kernels_stats.csv 2>/dev/null
cp gpurun_out/r06_new_kernels.jsonl profiles/r06_new_kernels.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 > profiles/r06_bench_driver.json 2> gpurun_out/bench_r06_stamped.err
cp bench_detail.json profiles/r06_bench_detail.json 2>/dev/null
mkdir -p gpurun_out/profiles_r06
cp profiles/r06_* profiles/isa_mix.json profiles/pmc_traffic.json gpurun_out/profiles_r06/
# the raw traces stay on the box: only the summaries and the per-kernel stats travel
find gpurun_out/prof -name "*kernel_trace.csv" -delete
find gpurun_out/prof -name "*counter_collection.csv" -size +2M -delete
du -sh gpurun_out; tail -c 3000 profiles/r06_bench_driver.json

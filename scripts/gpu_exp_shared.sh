#!/bin/bash
# timing experiment: k_shared_step with pieces compiled out (variant libraries under rsrl_amd/lib/variants)
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/expsh; rm -rf $O; mkdir -p $O
cd /tmp
for v in base noread nowrite noreaddrsrl_exp_nowrite; do
  if [ $v = base ]; then unset RSRL_HIP_LIB; else export RSRL_HIP_LIB=$GRAFT_REPO_ROOT/rsrl_amd/lib/variants/$v.so; fi
  python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier none | sed "s/^/$v /"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -o s -- python $GRAFT_REPO_ROOT/scripts/prof_shared.py fourier none > $O/$v.log 2>&1
  grep k_shared_step $O/$v/s_kernel_stats.csv | cut -d, -f1-5 | cut -c1-200
done

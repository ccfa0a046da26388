# after gpu_round6_fuzz_last.sh: the sparse-trace family alone and the long horizons (seeds 411-413)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_r06e
FUZZ_FAMILY=sparse_lambda timeout 500 python tests/fuzz_parity.py 800 411 > gpurun_out/fuzz_r06e/sparse.log 2>&1; echo "sparse rc=$?"; tail -1 gpurun_out/fuzz_r06e/sparse.log | cut -c1-300
FUZZ_LONG=1 timeout 600 python tests/fuzz_parity.py 200 412 > gpurun_out/fuzz_r06e/long.log 2>&1; echo "long rc=$?"; tail -1 gpurun_out/fuzz_r06e/long.log | cut -c1-300
FUZZ_FAMILY=sparse_lambda FUZZ_LONG=1 timeout 400 python tests/fuzz_parity.py 80 413 > gpurun_out/fuzz_r06e/sparse_long.log 2>&1; echo "sparse long rc=$?"; tail -1 gpurun_out/fuzz_r06e/sparse_long.log | cut -c1-300
for f in gpurun_out/fuzz_r06e/*.log; do grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -200 > $f.tail; mv $f.tail $f; done

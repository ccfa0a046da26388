# the round's last visit: a shorter differential campaign on the re-built final tree (seeds 401-405)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_r06d
for seed in 401 402; do timeout 900 python tests/fuzz_parity.py 2500 $seed > gpurun_out/fuzz_r06d/parity_$seed.log 2>&1; echo "parity $seed rc=$?"; tail -1 gpurun_out/fuzz_r06d/parity_$seed.log | cut -c1-300; done
timeout 700 python tests/fuzz_f64.py 2000 403 > gpurun_out/fuzz_r06d/f64.log 2>&1; echo "f64 rc=$?"; tail -1 gpurun_out/fuzz_r06d/f64.log | cut -c1-1800
timeout 400 python tests/fuzz_abi.py 2000 404 > gpurun_out/fuzz_r06d/abi.log 2>&1; echo "abi rc=$?"; tail -1 gpurun_out/fuzz_r06d/abi.log | cut -c1-300
timeout 600 python tests/fuzz_ranks.py 300 405 > gpurun_out/fuzz_r06d/ranks.log 2>&1; echo "ranks rc=$?"; tail -1 gpurun_out/fuzz_r06d/ranks.log | cut -c1-300
for f in gpurun_out/fuzz_r06d/*.log; do grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -300 > $f.tail; mv $f.tail $f; done

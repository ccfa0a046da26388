set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz_r06c
for seed in 301 302 303; do timeout 1800 python tests/fuzz_parity.py 3000 $seed > gpurun_out/fuzz_r06c/parity_$seed.log 2>&1; echo "parity $seed rc=$?"; tail -1 gpurun_out/fuzz_r06c/parity_$seed.log | cut -c1-300; done
timeout 1200 python tests/fuzz_f64.py 3000 304 > gpurun_out/fuzz_r06c/f64.log 2>&1; echo "f64 rc=$?"; tail -1 gpurun_out/fuzz_r06c/f64.log | cut -c1-1800
timeout 900 python tests/fuzz_abi.py 3000 305 > gpurun_out/fuzz_r06c/abi.log 2>&1; echo "abi rc=$?"; tail -1 gpurun_out/fuzz_r06c/abi.log | cut -c1-300
timeout 1200 python tests/fuzz_ranks.py 600 306 > gpurun_out/fuzz_r06c/ranks.log 2>&1; echo "ranks rc=$?"; tail -1 gpurun_out/fuzz_r06c/ranks.log | cut -c1-300
FUZZ_LONG=1 timeout 1200 python tests/fuzz_parity.py 300 307 > gpurun_out/fuzz_r06c/parity_long.log 2>&1; echo "long rc=$?"; tail -1 gpurun_out/fuzz_r06c/parity_long.log | cut -c1-300
for f in gpurun_out/fuzz_r06c/*.log; do grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $f | tail -300 > $f.tail; mv $f.tail $f; done

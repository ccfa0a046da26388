#!/bin/bash
# the round's closing visit on the final code: the GPU suite, smoke, the evidence pipeline (scripts/gpu_round6_profile.sh), two ranks on the one GPU under torchrun
set -u
cd $GRAFT_REPO_ROOT
bash scripts/gpu_tests.sh > gpurun_out/final_tests_tail.log 2>&1
grep -n "passed\|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/gpu_round6_profile.sh > gpurun_out/final_profile.log 2>&1
tail -c 1500 gpurun_out/final_profile.log
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --allow-oversubscribe > gpurun_out/final_two_ranks.json 2> gpurun_out/final_two_ranks.err
echo "two ranks rc=$?"; tail -n 1 gpurun_out/final_two_ranks.json | cut -c1-600

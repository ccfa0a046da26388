#!/bin/bash
# exercise the multi-rank code path on the 1-GPU box: torch imported first (its bundled HIP runtime), 2 ranks sharing GPU 0
set -u
mkdir -p gpurun_out
echo "== torch first, single rank"
timeout 600 python -c "
import torch, runpy, sys
sys.argv=['bench.py','--steps','2560','--warmup','256','--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>&1 | tail -3 | cut -c1-1500
echo "== torchrun 2 ranks on one GPU"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2560 --warmup 256 2>&1 | tail -5 | cut -c1-2500
echo "== plain"
timeout 600 python bench.py --steps 5120 --warmup 512 2>&1 | tail -1 | cut -c1-3000

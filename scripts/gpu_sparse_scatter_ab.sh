#!/bin/bash
# the sparse-trace scatter kernel at the sizes of DESIGN 4 (short lists: 8 warm-up steps; full lists: 1 000), learners per block by RSRL_SPARSE_CHUNK
cd $GRAFT_REPO_ROOT
for n in 8192 16384 65536 262144; do timeout 120 python scripts/sparse_scatter_time.py $n 8; done
timeout 120 python scripts/sparse_scatter_time.py 65536 1000
for ch in 512 1024 2048 4096; do RSRL_SPARSE_CHUNK=$ch timeout 120 python scripts/sparse_scatter_time.py 65536 8; done

#!/bin/bash
# A/B of builds of librsrl_hip.so on ONE GPU box (boxes differ by ~1 %): the driver's bench line for the in-tree build and for each
# alternative, twice round-robin, then the GPU tests on the LAST alternative.  Output: gpurun_out/ab.log
#   bash scripts/ab_lib.sh rsrl_amd/lib/<alt1>.so [rsrl_amd/lib/<alt2>.so ...]
mkdir -p gpurun_out; : > gpurun_out/ab.log
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(sys.argv[1], d['value'], d['roofline']['frac'])" "$1"; }
for rep in 1 2; do
  python bench.py 2>/dev/null | val in-tree >> gpurun_out/ab.log
  for a in "$@"; do RSRL_HIP_LIB=$PWD/$a python bench.py 2>/dev/null | val "$a" >> gpurun_out/ab.log; done
done
for a in "$@"; do last=$a; done
RSRL_HIP_LIB=$PWD/$last python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -5 >> gpurun_out/ab.log
cat gpurun_out/ab.log

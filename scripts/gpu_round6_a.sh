set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/profile_round5_kernels.py > gpurun_out/r06_new_kernels_a.jsonl 2> gpurun_out/r06_new_kernels_a.err
RSRL_LAMBDA_MEM1=1 python scripts/profile_round5_kernels.py 2>/dev/null | grep lambda_mem > gpurun_out/r06_lambda_mem1.jsonl
cat gpurun_out/r06_new_kernels_a.jsonl gpurun_out/r06_lambda_mem1.jsonl
python -c "
import ctypes as C
from rsrl_amd import _abi
o=C.c_double(); print('copy', _abi.lib().rsrl_hip_measure_copy(0, 1<<30, 10, C.byref(o)), o.value)
o=C.c_double(); print('copy 4G', _abi.lib().rsrl_hip_measure_copy(0, 1<<32, 5, C.byref(o)), o.value)"
bash scripts/gpu_tests.sh

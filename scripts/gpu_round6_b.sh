# round 6, second visit: bf16 + stochastic rounding for the wave family's trace / aux agents against f64 (teacher-forced), then the suites the change touches
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python - <<PY 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl"
import json, sys
sys.path.insert(0, "scripts")
import measure_parity as mp
out = {n: mp.teacher_forced(n) for n in ("w7_gq", "w7_td", "w7_sl")}
json.dump(out, open("gpurun_out/r06_parity_w7.json", "w"), indent=1)
for n, r in out.items():
    print(n, json.dumps({k: r[k] for k in ("f32", "bf16", "bf16_vs_f32", "max_abs_w_f64")}))
PY
timeout 1200 python -m pytest tests/test_gpu_parity_lambda.py tests/test_gpu_parity_gq.py tests/test_gpu_round4.py tests/test_gpu_parity_f64.py tests/test_gpu_fuzz.py -x -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -15

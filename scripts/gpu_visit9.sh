#!/bin/bash
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/v9
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "--- bench --gpus 2 (own spawn, 2 ranks on the one device)"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --allow-oversubscribe --no-cpu-baseline > $O/bench_spawn2.json 2> $O/bench_spawn2.err; echo "rc $?"; cut -c1-600 $O/bench_spawn2.json; tail -3 $O/bench_spawn2.err
python - <<PY
import json
for l in open("$O/bench_spawn2.json"):
    if l.startswith("{"):
        d=json.loads(l); print({k:d.get(k) for k in ("value","n_gpus","oversubscribed")}, d["config"]["ranks"], d.get("shared_w"), d.get("shared_w_peer"))
PY
echo "--- under torch.distributed.run (the driver's launcher), refused without the flag"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_tr2.json 2> $O/bench_tr2.err; echo "rc $?"; tail -2 $O/bench_tr2.err | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity_mc.py -m gpu -q -k "cpp" --timeout 300 2>&1 | tail -3

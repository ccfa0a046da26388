#!/bin/bash
# the driver's bench invocation three times in a row: wall time and the headline / secondary figures of each run
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for i in 1 2 3; do
  t0=$(date +%s.%N)
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/soak$i.json 2> gpurun_out/soak$i.err
  t1=$(date +%s.%N)
  python - <<PY
import json
d=[json.loads(l) for l in open("gpurun_out/soak$i.json") if l.startswith("{")][0]
g=lambda k: (d.get(k) or {}).get("value")
print("run $i wall %.1f s" % ($t1 - $t0), "value %.4g" % d["value"], "streaming frac %.3f" % d["roofline_streaming"]["frac"], "c3 %.3g c5 %.3g rccl %.3g peer %.3g nocoalesce %.3g" % (g("c3_shared_tiles"), g("c5_wave_bf16"), g("shared_w_rccl"), g("shared_w"), g("value_no_coalesce")))
PY
done

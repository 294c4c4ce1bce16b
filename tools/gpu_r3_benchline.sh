#!/bin/bash
# the cfg3 bench line as the driver runs it, with the committed traffic file of the same sources (traffic_stale must read false)
mkdir -p gpurun_out/final
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/final/bench_cfg3.log 2>&1
tail -1 gpurun_out/final/bench_cfg3.log | cut -c1-400
python - <<'PY'
import json
for l in open("gpurun_out/final/bench_cfg3.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print(d["ms_per_step"], d["value"], d["config"]["uninstrumented_ms_per_step"], r["achieved"], r["frac"], r["traffic"], r["traffic_stale"], r["step"])
PY

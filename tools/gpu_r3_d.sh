#!/bin/bash
# round 3, call D: weight gradient on a side stream -- bench with / without, then the whole GPU suite with it on
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for v in "side:" "noside:--no-wgrad-stream"; do
  tag=${v%%:*}; fl=${v#*:}
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline $fl > gpurun_out/r3d_bench_$tag.log 2>&1; echo "bench $tag rc=$?"
done
python - <<'PY'
import json
for tag in ("side", "noside"):
    for l in open("gpurun_out/r3d_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], "uninstrumented", d["config"].get("uninstrumented_ms_per_step"), r["kernel"], r["achieved"], r["frac"])
PY
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3d_tests_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3d_tests_gpu.log | tail -15 | cut -c1-300
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --accumulate 4 --batch 16 > gpurun_out/r3d_bench_acc4.log 2>&1; echo "bench acc rc=$?"; tail -1 gpurun_out/r3d_bench_acc4.log | cut -c1-400

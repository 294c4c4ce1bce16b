#!/bin/bash
# round 3, call H: validation -- the whole GPU suite (incl. the StyleGAN2 goldens at cfg4's real network), smoke, the default bench
# line (cfg3 with the CPU baseline leg), cfg4 with the FC layers on the HIP GEMM
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3h_tests_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3h_tests_gpu.log | tail -15 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3h_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r3h_smoke.log | cut -c1-300
timeout 600 python bench.py > gpurun_out/r3h_bench_cfg3.log 2>&1; echo "bench rc=$?"
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3h_bench_cfg4.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg4"):
    for l in open("gpurun_out/r3h_bench_%s.log" % w):
        if l.startswith("{"):
            d = json.loads(l); r = d.get("roofline") or {}
            print("BENCH", w, d["ms_per_step"], d["value"], "uninstr", (d.get("config") or {}).get("uninstrumented_ms_per_step"), json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac", "traffic", "traffic_stale")}), "cpu:", json.dumps(d.get("cpu_baseline"))[:300])
PY

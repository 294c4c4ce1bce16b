#!/bin/bash
# round 2, call C: full GPU suite (plugin / typed ops / IC-deep / snapshot / SN + dbias changes), smoke, bench lines of every workload
mkdir -p gpurun_out; rm -f gpurun_out/bench_shape_parity.txt
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_gpu.log | tail -45
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 300 python tools/hbm_bench.py > gpurun_out/hbm_bench.log 2>&1; grep -v amdgpu gpurun_out/hbm_bench.log | tail -30
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_cfg3.log 2>&1
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 > gpurun_out/bench_cfg5.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 > gpurun_out/bench_cfg4.log 2>&1
timeout 300 python bench.py --workload sample --steps 20 --warmup 3 > gpurun_out/bench_sample.log 2>&1
timeout 300 python bench.py --workload cfg2 --steps 10 --warmup 3 > gpurun_out/bench_cfg2.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg5", "cfg4", "sample", "cfg2"):
    f = "gpurun_out/bench_%s.log" % w
    ok = False
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); ok = True
            r = d.get("roofline") or {}
            k = r.pop("all_conv_kernels", {})
            print("BENCH", w, d["ms_per_step"], d["value"], "roof:", json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac", "step")}), "hbm:", json.dumps((d.get("roofline_hbm") or {}).get("all_hbm_ops")), "cpu:", json.dumps(d.get("cpu_baseline"))[:300])
            if w == "cfg3":
                print({n: (v["executed_tflops"], v["ms_per_step"]) for n, v in k.items() if "planes_kernel" in n or "gemm_kernel" in n})
    if not ok:
        print("BENCH", w, "NO JSON LINE"); print(open(f).read()[-1500:])
PY

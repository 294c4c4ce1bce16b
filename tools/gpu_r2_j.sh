#!/bin/bash
# round 2, call J: full GPU suite, smoke, every bench line, kernel trace + HBM-traffic + MFMA-utilisation passes of the cfg3 step
mkdir -p gpurun_out/prof; rm -f gpurun_out/bench_shape_parity.txt
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/tests_gpu.log | tail -25
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 400 python bench.py > gpurun_out/bench_cfg3.log 2>&1
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 > gpurun_out/bench_cfg5.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 > gpurun_out/bench_cfg4.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --fp16 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
timeout 300 python bench.py --workload sample --steps 20 --warmup 3 > gpurun_out/bench_sample.log 2>&1
timeout 300 python bench.py --workload cfg2 --steps 10 --warmup 3 > gpurun_out/bench_cfg2.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg5", "cfg4", "cfg4_fp16", "sample", "cfg2"):
    f = "gpurun_out/bench_%s.log" % w
    ok = False
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); ok = True
            r = d.get("roofline") or {}
            print("BENCH", w, d["ms_per_step"], d["value"], json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac", "traffic", "step")}), "cpu:", json.dumps(d.get("cpu_baseline"))[:200])
    if not ok:
        print("BENCH", w, "NO JSON LINE"); print(open(f).read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --init N02 > $R/gpurun_out/prof/rocprof_run.log 2>&1
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/r02_bench_cfg3_kernel_stats.csv \;
python - <<'PY'
import csv, os
p = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof/r02_bench_cfg3_kernel_stats.csv"
rows = list(csv.DictReader(open(p)))
steps = 5
print("kernel time per step %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
    print("%8.2f ms/step %6.1f calls/step %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
cd $R && ICG_PMC_STEPS=3 bash tools/gpu_pmc_hbm.sh > gpurun_out/pmc_hbm_run.log 2>&1; tail -c 400 gpurun_out/pmc_hbm_run.log
cd $R && bash tools/gpu_pmc_sq.sh 2>&1 | head -12

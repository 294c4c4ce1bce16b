#!/bin/bash
# round 2, call D: suite after the SN / fp16-block / kNN changes, cfg4 fp16 bench, kernel-stats profile of the cfg3 step
mkdir -p gpurun_out/prof; rm -f gpurun_out/bench_shape_parity.txt
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/tests_gpu.log | tail -45
timeout 300 python tools/hbm_bench.py 2>&1 | grep -E "sn_forward|copy" 
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --fp16 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg3.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg4_fp16"):
    f = "gpurun_out/bench_%s.log" % w
    ok = False
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); ok = True
            r = d.get("roofline") or {}
            r.pop("all_conv_kernels", {})
            print("BENCH", w, d["ms_per_step"], d["value"], json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac")}), "hbm:", json.dumps((d.get("roofline_hbm") or {}).get("all_hbm_ops")))
    if not ok:
        print("BENCH", w, "NO JSON LINE"); print(open(f).read()[-2500:])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --init N02 > $R/gpurun_out/prof/rocprof_run.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/prof/r02_bench_cfg3_kernel_stats.csv 2>/dev/null || find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/r02_bench_cfg3_kernel_stats.csv \;
python - <<'PY'
import csv, os
p = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof/r02_bench_cfg3_kernel_stats.csv"
rows = list(csv.DictReader(open(p)))
steps = 5
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("kernel time per step %.1f ms" % tot)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:42]:
    print("%8.2f ms/step %6d calls/step %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) // steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY

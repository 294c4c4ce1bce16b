#!/bin/bash
# round 3, call F: 96-row tiles in the weight-gradient plane GEMM (first contact) + A/B, HBM-traffic PMC passes of the cfg3 step,
# kernel trace, MFMA-utilisation pass
mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "plane_gemm" -p no:cacheprovider > gpurun_out/r3f_plane_tests.log 2>&1; echo "plane tests rc=$?"
tail -3 gpurun_out/r3f_plane_tests.log | cut -c1-300
timeout 300 python tools/pgemm_bench.py > gpurun_out/r3f_pgemm.log 2>&1
ICG_PGEMM_TN_WM3=0 timeout 300 python tools/pgemm_bench.py > gpurun_out/r3f_pgemm_wm4.log 2>&1
sed -n '/weight-gradient/,$p' gpurun_out/r3f_pgemm.log > /tmp/n.txt; sed -n '/weight-gradient/,$p' gpurun_out/r3f_pgemm_wm4.log > /tmp/o.txt; paste -d'\n' /tmp/n.txt /tmp/o.txt | cut -c1-150
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3f_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r3f_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH", d["ms_per_step"], d["value"], "uninstrumented", d["config"].get("uninstrumented_ms_per_step"), r["kernel"], r["achieved"], r["frac"])
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-uninstrumented-leg --init N02 > $R/gpurun_out/prof/rocprof_run.log 2>&1
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/r03_bench_cfg3_kernel_stats.csv \;
python - <<'PY'
import csv, os
p = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof/r03_bench_cfg3_kernel_stats.csv"
rows = list(csv.DictReader(open(p)))
steps = 5
print("kernel time per step %.1f ms, launches per step %.0f" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%8.2f ms/step %6.1f calls/step %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
cd $R && ICG_PMC_STEPS=3 bash tools/gpu_pmc_hbm.sh > gpurun_out/pmc_hbm_run.log 2>&1; tail -c 600 gpurun_out/pmc_hbm_run.log
cd $R && bash tools/gpu_pmc_sq.sh 2>&1 | head -14

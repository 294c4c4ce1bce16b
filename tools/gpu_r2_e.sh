#!/bin/bash
# round 2, call E: measurement evidence -- kernel trace of the cfg4 iteration, HBM-traffic PMC passes and the MFMA-utilisation
# PMC pass of the cfg3 step (each PMC pass in its own run, kernel trace only)
mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
export ICG_PMC_STEPS=3 ICG_PMC_COMMIT=${ICG_PMC_COMMIT:-unknown}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof4
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python $R/bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_cfg4.log 2>&1
find /tmp/prof4 -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof/r02_bench_cfg4_kernel_stats.csv \;
python - <<'PY'
import csv, os
p = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof/r02_bench_cfg4_kernel_stats.csv"
rows = list(csv.DictReader(open(p)))
steps = 20
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print("cfg4 kernel time per iteration %.1f ms, launches per iteration %d" % (tot, sum(int(r["Calls"]) for r in rows) // steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print("%8.2f ms/it %6.1f calls/it %9.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
grep '^{' $R/gpurun_out/prof/rocprof_cfg4.log | cut -c1-400
cd $R && bash tools/gpu_pmc_hbm.sh > gpurun_out/pmc_hbm_run.log 2>&1; tail -c 700 gpurun_out/pmc_hbm_run.log
cd $R && bash tools/gpu_pmc_sq.sh 2>&1 | head -30

#!/bin/bash
# round 3, final evidence (a): the cfg3 bench line as the driver runs it (with the CPU baseline), its rocprofv3 kernel trace, the secondary
# workloads.  Outputs under gpurun_out/final/ (copied to profiles/r03_*).
mkdir -p gpurun_out/final
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_cfg3.log 2>&1
timeout 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg2.log 2>&1
timeout 600 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.log 2>&1
timeout 300 python bench.py --workload sample --no-cpu-baseline > $O/bench_sample.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_cfg4.log 2>&1
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > $O/bench_cfg4_fp16.log 2>&1
timeout 300 python bench.py --accumulate 4 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_acc4x16.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof3 /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof3 -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-uninstrumented-leg > $O/rocprof_cfg3.log 2>&1
cp /tmp/prof3/bench_kernel_stats.csv $O/bench_cfg3_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o cfg4 -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timer > $O/rocprof_cfg4_fp16.log 2>&1
cp /tmp/prof4/cfg4_kernel_stats.csv $O/bench_cfg4_fp16_kernel_stats.csv
cd $R
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.log")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d.get("roofline") or {}
            print(f.split("/")[-1], d["ms_per_step"], d["value"], d["config"].get("uninstrumented_ms_per_step"), r.get("kernel"), r.get("achieved"), r.get("frac"), r.get("traffic_stale"))
PY
head -12 $O/bench_cfg3_kernel_stats.csv | cut -c1-140

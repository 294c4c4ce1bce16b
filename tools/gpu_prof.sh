#!/bin/bash
mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_run.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/prof/
tail -n 1 $R/gpurun_out/prof/rocprof_run.log | cut -c1-300

#!/bin/bash
# full GPU validation: all gpu tests, smoke, bench (with cpu baseline), rocprof kernel stats -> gpurun_out/
mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1
echo "tests exit $?" >> gpurun_out/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 1200 python bench.py > gpurun_out/bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_run.log 2>&1
cp /tmp/prof/bench_kernel_stats.csv $R/gpurun_out/prof/ 2>/dev/null
cd $R
tail -n 3 gpurun_out/tests_gpu.log; tail -n 2 gpurun_out/smoke.log; tail -n 1 gpurun_out/bench.log | cut -c1-2500

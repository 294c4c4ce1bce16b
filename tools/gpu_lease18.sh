#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_decision_replay_gpu.py -m gpu -q -x > gpurun_out/l18_tests.log 2>&1
echo "tests exit $?"; tail -n 4 gpurun_out/l18_tests.log | cut -c1-250
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_cfg4.log 2>&1
cp /tmp/prof4/bench_kernel_stats.csv $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv
tail -n 1 $R/gpurun_out/prof/rocprof_cfg4.log | cut -c1-330

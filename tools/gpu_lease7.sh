#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_decision_replay_gpu.py tests/test_sg2_fused_gpu.py -m gpu -q > gpurun_out/l7_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/l7_tests.log | cut -c1-250
cat gpurun_out/decision_replay_cfg3_w96_r256_b16.txt
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_cfg4.log 2>&1
cp /tmp/prof4/bench_kernel_stats.csv $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv
head -n 12 $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv | cut -c1-140

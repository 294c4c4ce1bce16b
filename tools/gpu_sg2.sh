#!/bin/bash
# StyleGAN2 (cfg4) evidence in one lease: fused-layer kernel parity, network goldens, A/B bench fused vs composed, kernel stats
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out/prof
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py tests/test_stylegan_conv.py tests/test_stylegan_ops.py -m gpu -x -q > gpurun_out/sg2_tests.log 2>&1
echo "tests exit $?"; tail -n 25 gpurun_out/sg2_tests.log | cut -c1-300
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16.log | cut -c1-330
ICG_SG2_FUSED=0 timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16_composed.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16_composed.log | cut -c1-330
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4.log 2>&1
tail -n 1 gpurun_out/bench_cfg4.log | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o bench -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > $R/gpurun_out/prof/rocprof_cfg4.log 2>&1
cp /tmp/prof4/bench_kernel_stats.csv $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv
head -n 30 $R/gpurun_out/prof/bench_cfg4_fp16_kernel_stats.csv | cut -c1-160

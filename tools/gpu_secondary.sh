#!/bin/bash
# secondary workloads + microbenchmarks + accuracy table -> gpurun_out/ (copy to profiles/)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/bench_cfg4.log 2>&1
timeout 900 python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2>&1
timeout 600 python bench.py --workload sample --no-cpu-baseline > gpurun_out/bench_sample.log 2>&1
timeout 600 python bench.py --workload sample --batch 1 --no-cpu-baseline > gpurun_out/bench_sample_b1.log 2>&1
timeout 900 python bench.py --no-winograd --no-cpu-baseline > gpurun_out/bench_cfg3_no_winograd.log 2>&1
timeout 600 python tools/wino_bench.py > gpurun_out/wino_bench.log 2>&1
timeout 600 python tools/rs_wino_bench.py > gpurun_out/rs_wino_bench.log 2>&1
timeout 600 python tests/diag_winograd_accuracy.py > gpurun_out/winograd_accuracy.log 2>&1
for f in bench_cfg4 bench_cfg5 bench_sample bench_sample_b1 bench_cfg3_no_winograd; do tail -n 1 gpurun_out/$f.log | cut -c1-260; done
grep -c "F(4,3)" gpurun_out/wino_bench.log; tail -n 3 gpurun_out/rs_wino_bench.log | cut -c1-200; tail -n 3 gpurun_out/winograd_accuracy.log | cut -c1-200

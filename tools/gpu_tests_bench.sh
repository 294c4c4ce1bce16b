#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1
echo "tests exit $?" >> gpurun_out/tests_gpu.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_quick.log 2>&1
grep -E "^E  |passed|failed|FAILED" gpurun_out/tests_gpu.log | head -20; tail -n 1 gpurun_out/bench_quick.log | cut -c1-700

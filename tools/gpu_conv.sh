#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "conv2d or gemm" > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench.log 2>&1
tail -n 4 gpurun_out/tests.log; cat gpurun_out/conv_bench.log

#!/bin/bash
# the dense-layer cases added to the kernel suite after the final run
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "test_conv2d_fprop or dgrad_as_fprop or wgrad" > gpurun_out/r3w_kern.log 2>&1; echo "rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3w_kern.log | tail -8 | cut -c1-300

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_stylegan_ops.py tests/test_stylegan_conv.py tests/test_stylegan2.py tests/test_sg2_fused_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/l25_tests.log 2>&1
echo "tests exit $?"; tail -n 6 gpurun_out/l25_tests.log | cut -c1-250

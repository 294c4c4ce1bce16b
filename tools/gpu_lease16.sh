#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/sg2_greg_compare.py 2>&1 | grep -v amdgpu.ids > gpurun_out/sg2_greg_compare.txt; tail -n 30 gpurun_out/sg2_greg_compare.txt | cut -c1-200
timeout 600 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py -m gpu -q > gpurun_out/l16_tests.log 2>&1
echo "tests exit $?"; tail -n 8 gpurun_out/l16_tests.log | cut -c1-250

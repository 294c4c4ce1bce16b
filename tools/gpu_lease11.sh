#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py -m gpu -q -x > gpurun_out/l11_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/l11_tests.log | cut -c1-250
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16.log | cut -c1-330
timeout 600 python tools/hbm_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/hbm_bench.txt; grep "sg2\|upfirdn2d_nhwc fp16\|ceiling" gpurun_out/hbm_bench.txt | cut -c1-170

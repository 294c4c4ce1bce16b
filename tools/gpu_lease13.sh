#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py -m gpu -q -x > gpurun_out/l13_tests.log 2>&1
echo "tests exit $?"; tail -n 8 gpurun_out/l13_tests.log | cut -c1-250
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16.log | cut -c1-330
ICG_SG2_FUSED2=0 timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16_composed_reg.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16_composed_reg.log | cut -c1-330
timeout 300 python tools/sg2_phase_times.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sg2_phase_times.txt | tail -4

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/sg2_route_check.txt
timeout 600 python -m pytest tests/test_sg2_fused_gpu.py tests/test_stylegan2.py tests/test_decision_replay_gpu.py -m gpu -x -q > gpurun_out/l5_tests.log 2>&1
echo "tests exit $?"; tail -n 12 gpurun_out/l5_tests.log | cut -c1-250
cat gpurun_out/sg2_route_check.txt
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
tail -n 1 gpurun_out/bench_cfg4_fp16.log | cut -c1-330
timeout 900 python tools/parity_report.py --stats > gpurun_out/parity_report.txt 2>&1
grep -c PARITY gpurun_out/parity_report.txt; grep "PARITY" gpurun_out/parity_report.txt | cut -c1-260 | head -40

#!/bin/bash
# last check of the round: the full GPU suite and smoke() at the final commit
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1
echo "tests exit $?" >> gpurun_out/tests_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke exit $?" >> gpurun_out/smoke.log
grep -E "^(FAILED|ERROR)|passed|failed|exit" gpurun_out/tests_gpu.log | tail -5 | cut -c1-250; tail -2 gpurun_out/smoke.log | cut -c1-200

#!/bin/bash
# full GPU suite + timing of the slowest tests -> gpurun_out/full_tests.log
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
rm -f gpurun_out/sg2_route_check.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/full_tests.log 2>&1
echo "tests exit $?"; tail -n 40 gpurun_out/full_tests.log | cut -c1-250
cat gpurun_out/sg2_route_check.txt

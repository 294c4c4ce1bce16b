#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -q --tb=short -p no:cacheprovider -x > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_cfg3.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof | head -20
tail -n 3 gpurun_out/tests.log; tail -n 1 gpurun_out/bench_cfg3.log | cut -c1-1500

#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof; find /tmp/prof -name "*stats*" -exec cp {} gpurun_out/prof/ \; ; ls -la /tmp/prof/* | head; ls -la gpurun_out/prof
cat gpurun_out/conv_bench.log

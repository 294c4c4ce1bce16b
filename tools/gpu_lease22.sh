#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_decision_replay_gpu.py -m gpu -q -x -k "linear_group or sn_ or train or step or decisions" -p no:cacheprovider > gpurun_out/l22_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/l22_tests.log | cut -c1-250
for v in 0 1 0 1 0 1; do
  ICG_CCBN_GROUP=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('CCBN_GROUP=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', r['frac'])"
done
bash tools/gpu_launch_count.sh | head -3 | cut -c1-160

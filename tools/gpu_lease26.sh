#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_decision_replay_gpu.py -m gpu -q -k "wgrad or train or step or decisions or attn or linear_group" -p no:cacheprovider > gpurun_out/l26_tests.log 2>&1
echo "tests exit $?"; tail -n 6 gpurun_out/l26_tests.log | cut -c1-250
for v in 0 1 0 1; do
  ICG_WGRAD_1X1_TN=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('WGRAD_1X1_TN=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', r['frac'], 'peak', d['config']['peak_hbm_gib'])"
done

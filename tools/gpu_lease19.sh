#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_decision_replay_gpu.py tests/test_ddp_rccl_gpu.py -m gpu -q -x -k "sn_ or bn_ or train or step or decisions or ddp or rccl or bench" -p no:cacheprovider > gpurun_out/l19_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/l19_tests.log | cut -c1-250
for v in 0 8 0 8; do
  ICG_SN_BWD_GROUP=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('SN_BWD_GROUP=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done
bash tools/gpu_launch_count.sh | head -12 | cut -c1-160

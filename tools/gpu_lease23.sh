#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -k "linear_group or sn_ or train_steps_vs_golden" -p no:cacheprovider > gpurun_out/l23_tests.log 2>&1
echo "tests exit $?"; tail -n 6 gpurun_out/l23_tests.log | cut -c1-250
timeout 300 python bench.py --steps 20 --warmup 5 --init N02 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('cfg3 ms_per_step', d['ms_per_step'], 'uninstr', d['config']['uninstrumented_ms_per_step'], 'peak GiB', d['config']['peak_hbm_gib'])"

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "attn or forward or train or step or gamma" -p no:cacheprovider > gpurun_out/l17_tests.log 2>&1
echo "tests exit $?"; tail -n 5 gpurun_out/l17_tests.log | cut -c1-250
for v in 0 1; do
  for w in G D; do
    ICG_ATTN_PROJ=$v ICG_ATTN_OUT=$v timeout 200 python tools/attn_block_prof.py $w 2>&1 | grep "attention block" | sed "s/^/ATTN_PROJ=$v /"
  done
done
for v in 0 1 0 1; do
  ICG_ATTN_PROJ=$v ICG_ATTN_OUT=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ATTN_PROJ=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done

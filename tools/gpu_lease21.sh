#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for v in 8 0 8 0 8 0 8 0 8 0; do
  ICG_SN_BWD_GROUP=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('SN_BWD_GROUP=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', r['frac'])"
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for v in 0 1 0 1 0 1 0 1; do
  ICG_WGRAD_1X1_TN=$v timeout 300 python bench.py --steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg 2>/dev/null | tail -n 1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; k=r['all_conv_kernels']
t=sum(v['ms_per_step'] for n,v in k.items() if ('icg_gemm_kernel<1, 1' in n or 'icg_pgemm_tn_kernel<' in n) and 'composite' not in n)
print('WGRAD_1X1_TN=$v cfg3 ms_per_step', d['ms_per_step'], 'frac', r['frac'], ' direct TN kernels ms/step %.2f' % t)"
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for e in "X=1" "ICG_CCBN_GROUP=0" "ICG_ATTN_PROJ=0" "ICG_ATTN_OUT=0" "ICG_SN_BWD_GROUP=0" "ICG_CCBN_GROUP=0 ICG_ATTN_PROJ=0 ICG_ATTN_OUT=0 ICG_SN_BWD_GROUP=0"; do
  env $e timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "test_train_steps_vs_golden and ic_r64_acc2-4" -p no:cacheprovider 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | sed "s/^/[$e] /"
done

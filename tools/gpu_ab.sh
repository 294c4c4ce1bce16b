#!/bin/bash
# A/B legs of one switch inside ONE gpurun lease (boxes differ by ~1 %, so variants are only ever compared inside a lease):
#   bash tools/gpu_ab.sh ICG_CCBN_GROUP "0 1 0 1"                       cfg3 step, 10 timed steps per leg
#   bash tools/gpu_ab.sh ICG_FIR_TILE "0 1 0 1" --workload cfg4 --fp16  any other bench.py workload
#   TESTS='-k "attn or train"' bash tools/gpu_ab.sh ...                 a pytest -m gpu selection first (stops the lease on failure)
# The round-5 records made this way: profiles/r05_cfg3_step_ab.txt, r05_pgemm_mfma32.txt, r05_cfg4_fused_layers_ab.txt.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
VAR=$1; VALUES=$2; shift 2
if [ -n "$TESTS" ]; then
  eval timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider $TESTS > gpurun_out/ab_tests.log 2>&1 || { tail -n 15 gpurun_out/ab_tests.log | cut -c1-250; exit 1; }
  tail -n 2 gpurun_out/ab_tests.log | cut -c1-200
fi
ARGS="$*"
case "$ARGS" in *--workload*) EXTRA="--steps 16 --warmup 4 --no-cpu-baseline" ;; *) EXTRA="--steps 10 --warmup 3 --init N02 --no-cpu-baseline --no-uninstrumented-leg" ;; esac
for v in $VALUES; do
  env $VAR=$v timeout 400 python bench.py $ARGS $EXTRA 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d.get('roofline') or {}
print('$VAR=$v', 'ms_per_step', d['ms_per_step'], 'frac', r.get('frac'), 'peak_hbm_gib', d['config'].get('peak_hbm_gib'))"
done

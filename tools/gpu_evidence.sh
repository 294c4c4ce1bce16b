#!/bin/bash
# The round's evidence set in one box, in dependency order: HBM PMC passes (the traffic file the bench line cites) -> MFMA PMC pass ->
# kernel stats -> the kept bench line (with the CPU baseline) -> per-block profile -> smoke.  Outputs under gpurun_out/; copy to profiles/rNN_*.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R && mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 200 python -c 'import __graft_entry__ as g; g.smoke()' > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" | tee -a gpurun_out/smoke.log
bash tools/gpu_pmc_hbm.sh > /dev/null 2>&1
cd $R
[ -s gpurun_out/hbm_traffic.json ] && cp gpurun_out/hbm_traffic.json "$(ls profiles/r*_hbm_traffic.json | tail -1)"
bash tools/gpu_pmc_sq.sh > /dev/null 2>&1
cd $R
bash tools/gpu_prof.sh
cd $R
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_final.log 2>&1
tail -n 1 gpurun_out/bench_final.log | cut -c1-3000
timeout 300 python tools/block_profile.py > gpurun_out/block_profile.txt 2>&1
tail -n 30 gpurun_out/block_profile.txt | cut -c1-200
if [ "$1" = "secondary" ]; then
  # the secondary workloads' lines at the same kernels (profiles/rNN_bench_{cfg2,cfg4_fp16,cfg5,sample,cfg3_acc4x16}.json.log)
  timeout 300 python bench.py --workload cfg2 --no-cpu-baseline > gpurun_out/bench_cfg2.log 2>&1
  timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/bench_cfg4_fp16.log 2>&1
  timeout 300 python bench.py --workload sample --no-cpu-baseline > gpurun_out/bench_sample.log 2>&1
  timeout 300 python bench.py --accumulate 4 --batch 16 --no-cpu-baseline > gpurun_out/bench_cfg3_acc4x16.log 2>&1
  timeout 400 python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/bench_cfg5.log 2>&1
  for f in cfg2 cfg4_fp16 sample cfg3_acc4x16 cfg5; do tail -n 1 gpurun_out/bench_$f.log | cut -c1-200; done
fi

#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q --tb=short -p no:cacheprovider -x -k "conv2d or gemm" > gpurun_out/tests.log 2>&1
echo "tests exit $?" >> gpurun_out/tests.log
timeout 600 python tools/conv_bench.py > gpurun_out/conv_bench_swz.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/bench_swz.log 2>&1
tail -n 3 gpurun_out/tests.log; cut -c1-130 gpurun_out/conv_bench_swz.log; tail -n1 gpurun_out/bench_swz.log | cut -c1-300
bash tools/gpu_pmc_hbm.sh > /dev/null 2>&1; python - <<PY
import json
d=json.load(open("gpurun_out/hbm_traffic.json"))
for k,v in d["kernels"].items():
    if "icg_gemm" in k and v["hbm_bytes_per_launch"]>2e8: print(k, v)
PY

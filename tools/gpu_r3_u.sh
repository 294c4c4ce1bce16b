#!/bin/bash
# round 3, call U: BigGAN's conditional-BN projections (64 rows x K = 657) on the row-streaming GEMM -- kernel tests, parity, A/B
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "conv2d or gemm_batched or linear" > gpurun_out/r3u_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3u_kern.log | tail -8 | cut -c1-300
true
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3u_bench_new.log 2>&1
python - <<'PY'
import json
for tag in ("new",):
    for l in open("gpurun_out/r3u_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH skinny", tag, d["ms_per_step"], d["value"], "uninstr", d["config"]["uninstrumented_ms_per_step"])
            for k, v in r["all_conv_kernels"].items():
                if "skinny" in k or "smallm" in k: print("   %7.2f ms %5d  %s" % (v["ms_per_step"], v["launches_per_step"], k))
PY
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r3u_parity.log 2>&1; echo "parity rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3u_parity.log | tail -8 | cut -c1-300

#!/bin/bash
# round 2, call G: plain-GEMM loader of the plane kernels; single-level chains for short-K plane GEMMs (threshold sweep)
# (historical: ICG_PLANES_1LEVEL_MAX_K / ICG_PLANES_RUN_KTILES were environment overrides when this ran; they are compile-time
# macros now -- tools/build_dbg.sh L1_384 builds the K <= 384 variant)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -k "wino or winograd or gemm or bench" -p no:cacheprovider 2>&1 | tail -4
for K in 0 192 384; do
  echo "== ICG_PLANES_1LEVEL_MAX_K=$K"
  ICG_PLANES_1LEVEL_MAX_K=$K timeout 300 python tools/parity_report.py cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 2>&1 | grep PARITY | grep "wino= 0" | cut -c1-330
  ICG_PLANES_1LEVEL_MAX_K=$K timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1l_$K.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_1l_$K.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH maxK=$K", d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
        for k, v in r["all_conv_kernels"].items():
            if "planes" in k and not k.startswith("composite"):
                print("   ", k, v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
PY
done
ICG_PLANES_1LEVEL_MAX_K=384 timeout 300 python tools/wino_bench.py 2>&1 | grep -v "^NOBLK" | head -30

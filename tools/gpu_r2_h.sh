#!/bin/bash
# round 2, call H: paired B-fragment LDS reads, packed flush adds, plain loaders for the weight-gradient plane GEMMs
# (historical: ICG_PLANES_1LEVEL_MAX_K / ICG_PLANES_RUN_KTILES were environment overrides when this ran; they are compile-time
# macros now -- tools/build_dbg.sh L1_384 builds the K <= 384 variant)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py tests/test_stylegan_conv.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/parity_report.py cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 cc_ic_r64 2>&1 | grep PARITY | cut -c1-300
for K in 0 384; do
  ICG_PLANES_1LEVEL_MAX_K=$K timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_h_$K.log 2>&1
  python - <<PY
import json
for l in open("gpurun_out/bench_h_$K.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH maxK=$K", d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
        for k, v in r["all_conv_kernels"].items():
            if not k.startswith("composite") and v["ms_per_step"] > 2:
                print("   ", k, v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
PY
done

#!/bin/bash
# round 2, call I: persistent plane GEMM (runs of output tiles per workgroup) vs one tile per workgroup
# (historical: ICG_PLANES_1LEVEL_MAX_K / ICG_PLANES_RUN_KTILES were environment overrides when this ran; they are compile-time
# macros now -- tools/build_dbg.sh L1_384 builds the K <= 384 variant)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "wino or winograd or bench" -p no:cacheprovider 2>&1 | tail -6
for V in product NOPERSIST; do
  for K in 0 384; do
    if [ $V = product ]; then
      ICG_PLANES_1LEVEL_MAX_K=$K timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_i_${V}_$K.log 2>&1
    else
      ICG_PLANES_1LEVEL_MAX_K=$K timeout 300 python tools/run_with_lib.py tools/libdbg_$V.so bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/bench_i_${V}_$K.log 2>&1
    fi
    python - <<PY
import json
ok = False
for l in open("gpurun_out/bench_i_${V}_$K.log"):
    if l.startswith("{"):
        ok = True
        d = json.loads(l); r = d["roofline"]
        print("BENCH $V maxK=$K", d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
        for k, v in r["all_conv_kernels"].items():
            if "planes" in k and not k.startswith("composite") and "<0, 0" in k:
                print("   ", k, v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
if not ok:
    print(open("gpurun_out/bench_i_${V}_$K.log").read()[-1500:])
PY
  done
done
timeout 200 python tools/parity_report.py cfg1_icgan_res64 cfg3_w96_r256 2>&1 | grep PARITY | cut -c1-200

#!/bin/bash
# round 3, call E: second-generation implicit-GEMM convolution (icg_pconv_kernel), first contact: every convolution kernel test with
# the size threshold removed (all eligible shapes on the new kernel), then at the product threshold, bench-shape parity, bench A/B
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
ICG_PCONV_MIN_TILES=1 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_stylegan_conv.py -q -p no:cacheprovider > gpurun_out/r3e_kernels_min1.log 2>&1; echo "kernel tests (threshold 1) rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3e_kernels_min1.log | tail -12 | cut -c1-300
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -p no:cacheprovider > gpurun_out/r3e_kernels.log 2>&1; echo "kernel + bench-shape tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3e_kernels.log | tail -12 | cut -c1-300
for v in "pconv:" "nopconv:ICG_PCONV=0"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3e_bench_$tag.log 2>&1; echo "bench $tag rc=$?"
done
python - <<'PY'
import json
for tag in ("pconv", "nopconv"):
    for l in open("gpurun_out/r3e_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], "uninstrumented", d["config"].get("uninstrumented_ms_per_step"), r["kernel"], r["achieved"], r["frac"])
            for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
                if not k.startswith("composite") and v["ms_per_step"] > 0.4:
                    print("   %7.2f ms %6.1f TF %3d  %s" % (v["ms_per_step"], v["executed_tflops"], v["launches_per_step"], k[:110]))
PY
timeout 900 python -m pytest tests/test_parity_gpu.py -q -p no:cacheprovider > gpurun_out/r3e_parity.log 2>&1; echo "parity rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3e_parity.log | tail -12 | cut -c1-300

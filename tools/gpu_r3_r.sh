#!/bin/bash
# round 3, call R: phased zero-inserted fp16 convolution -- kernel tests, StyleGAN2 suites, cfg4 fp16 bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "f16" > gpurun_out/r3r_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3r_kern.log | tail -15 | cut -c1-400
timeout 900 python -m pytest tests/test_stylegan_conv.py tests/test_stylegan2.py -m gpu -q -p no:cacheprovider > gpurun_out/r3r_sg2.log 2>&1; echo "sg2 tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3r_sg2.log | tail -15 | cut -c1-400
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3r_cfg4_fp16.log 2>&1
python - <<'PY'
import json
for tag in ("cfg4_fp16",):
    for l in open("gpurun_out/r3r_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
            for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
                print("   %7.2f ms %5d  %6.1f TF  %s" % (v["ms_per_step"], v["launches_per_step"], v["executed_tflops"], k[:100]))
PY

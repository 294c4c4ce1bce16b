#!/bin/bash
# round 3, call L: DBlock shortcut in the gradient chain (icg_avgpool2_bwd_add), stratified-sampling kernel timer, per-block profile
# of the step, resample-fused / stride-1 Winograd thresholds against the second-generation implicit GEMM
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_biggan_deep.py -m gpu -q -p no:cacheprovider -k "pool or golden or deep or train" > gpurun_out/r3l_tests.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3l_tests.log | tail -10 | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3l_bench_p4.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timer-period 1 --no-uninstrumented-leg > gpurun_out/r3l_bench_p1.log 2>&1
python - <<'PY'
import json
for tag in ("p4", "p1"):
    for l in open("gpurun_out/r3l_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], "uninstr", d["config"]["uninstrumented_ms_per_step"], r["kernel"], r["achieved"], r["frac"], r["launches"], r["avg_launch_ms"])
            for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
                print("   %7.2f ms %5d  %6.1f TF  %s" % (v["ms_per_step"], v["launches_per_step"], v["executed_tflops"], k[:100]))
PY
timeout 300 python tools/block_profile.py --steps 3 > gpurun_out/r3l_block_profile.txt 2>&1; tail -40 gpurun_out/r3l_block_profile.txt
timeout 300 python tools/rs_wino_bench.py > gpurun_out/r3l_rs_wino.txt 2>&1; cat gpurun_out/r3l_rs_wino.txt
WINO_BENCH_FLAGS=1 timeout 300 python tools/wino_bench.py > gpurun_out/r3l_wino_relu.txt 2>&1; grep -v SG2 gpurun_out/r3l_wino_relu.txt

#!/bin/bash
# round 3, call S: single-level chains in the forward / data-gradient plane GEMMs from K >= 384 (ICG_PGEMM_L1_MINK) -- parity suites + A/B bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3s_bench_l2.log 2>&1
ICG_PGEMM_L1_MINK=384 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3s_bench_l1_384.log 2>&1
ICG_PGEMM_L1_MINK=768 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3s_bench_l1_768.log 2>&1
python - <<'PY'
import json
for tag in ("l2", "l1_384", "l1_768"):
    for l in open("gpurun_out/r3s_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], "uninstr", d["config"]["uninstrumented_ms_per_step"], r["kernel"], r["achieved"], r["frac"])
            for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:5]:
                print("   %7.2f ms %5d  %6.1f TF  %s" % (v["ms_per_step"], v["launches_per_step"], v["executed_tflops"], k[:100]))
PY
ICG_PGEMM_L1_MINK=384 timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r3s_parity_384.log 2>&1; echo "parity(384) rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3s_parity_384.log | tail -12 | cut -c1-300
ICG_PGEMM_L1_MINK=384 timeout 600 python tools/parity_report.py --stats > gpurun_out/r3s_parity_report_384.txt 2>&1; tail -30 gpurun_out/r3s_parity_report_384.txt | cut -c1-220

#!/bin/bash
# round 3, call K: gradient chains (attention projections, GBlock shortcut: the consumers of x add the running gradient in their
# data-gradient epilogues) -- parity suite, StyleGAN2 real-network tests, BigGAN-deep, bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_checkpoint.py tests/test_biggan_deep.py tests/test_stylegan2.py tests/test_ddp_rccl_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r3k_tests.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3k_tests.log | tail -10 | cut -c1-300
for tag in a b; do timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3k_bench_$tag.log 2>&1; done
python - <<'PY'
import json
for tag in ("a", "b"):
    for l in open("gpurun_out/r3k_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], "uninstr", d["config"]["uninstrumented_ms_per_step"], r["kernel"], r["achieved"], r["frac"])
PY

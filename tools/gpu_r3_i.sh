#!/bin/bash
# round 3, call I: vectorised split-K reductions, StyleGAN2 dense layers on icg_gemm_batched; kernel + StyleGAN2 tests, cfg3 / cfg4 bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_stylegan2.py tests/test_stylegan_conv.py tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r3i_tests.log 2>&1; echo "tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3i_tests.log | tail -10 | cut -c1-300
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3i_bench_cfg3.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3i_bench_cfg4.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --fp16 --no-cpu-baseline > gpurun_out/r3i_bench_cfg4_fp16.log 2>&1
python - <<'PY'
import json
for w in ("cfg3", "cfg4", "cfg4_fp16"):
    for l in open("gpurun_out/r3i_bench_%s.log" % w):
        if l.startswith("{"):
            d = json.loads(l); r = d.get("roofline") or {}
            print("BENCH", w, d["ms_per_step"], d["value"], "uninstr", (d.get("config") or {}).get("uninstrumented_ms_per_step"), json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac")}))
PY

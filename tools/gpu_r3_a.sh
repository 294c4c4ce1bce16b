#!/bin/bash
# round 3, call A: first contact of the second-generation plane GEMM (csrc/pgemm.hip) with the hardware --
# its kernel tests, the microbenchmark on both generations, the Winograd kernel tests, the HBM microbench regeneration
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "plane_gemm" -p no:cacheprovider > gpurun_out/r3a_plane_tests.log 2>&1; echo "plane tests rc=$?"
tail -5 gpurun_out/r3a_plane_tests.log
timeout 300 python tools/pgemm_bench.py > gpurun_out/r3a_pgemm_new.log 2>&1; echo "bench new rc=$?"
ICG_PGEMM=0 timeout 300 python tools/pgemm_bench.py > gpurun_out/r3a_pgemm_old.log 2>&1; echo "bench old rc=$?"
paste -d'\n' gpurun_out/r3a_pgemm_new.log gpurun_out/r3a_pgemm_old.log | cut -c1-150
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -q -p no:cacheprovider > gpurun_out/r3a_kernel_tests.log 2>&1; echo "kernel tests rc=$?"
tail -4 gpurun_out/r3a_kernel_tests.log
timeout 900 python -m pytest tests/test_parity_gpu.py -q -p no:cacheprovider > gpurun_out/r3a_parity.log 2>&1; echo "parity rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3a_parity.log | tail -15
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3a_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r3a_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH", d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
        for k, v in r["all_conv_kernels"].items():
            if "planes" in k or "pgemm" in k:
                print("   ", k[:120], v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
PY
timeout 300 python tools/hbm_bench.py > gpurun_out/r3a_hbm_bench.log 2>&1; tail -30 gpurun_out/r3a_hbm_bench.log

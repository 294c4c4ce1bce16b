#!/bin/bash
# round 3, call B: weight-gradient plane GEMM of the second generation (first contact), both microbenchmarks on both
# generations, parity report with error statistics against the fp64 reference, the whole GPU suite, the cfg3 bench line
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "plane_gemm" -p no:cacheprovider > gpurun_out/r3b_plane_tests.log 2>&1; echo "plane tests rc=$?"
tail -5 gpurun_out/r3b_plane_tests.log | cut -c1-300
timeout 400 python tools/pgemm_bench.py > gpurun_out/r3b_pgemm_new.log 2>&1; echo "bench new rc=$?"
ICG_PGEMM=0 timeout 400 python tools/pgemm_bench.py > gpurun_out/r3b_pgemm_old.log 2>&1; echo "bench old rc=$?"
paste -d'\n' gpurun_out/r3b_pgemm_new.log gpurun_out/r3b_pgemm_old.log | grep -v "^G\.\|^D\.b[234]" | cut -c1-150
sed -n '/weight-gradient/,$p' gpurun_out/r3b_pgemm_new.log > /tmp/n.txt; sed -n '/weight-gradient/,$p' gpurun_out/r3b_pgemm_old.log > /tmp/o.txt; paste -d'\n' /tmp/n.txt /tmp/o.txt | cut -c1-150
timeout 900 python tools/parity_report.py --stats cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 cfg3_w96_r256_b16 cfg3_w96_r256_b64 > gpurun_out/r3b_parity_report.log 2>&1; echo "parity report rc=$?"
grep -E "PARITY|STATS|all [0-9]+ tensors" gpurun_out/r3b_parity_report.log | cut -c1-400
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3b_tests_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3b_tests_gpu.log | tail -15 | cut -c1-300
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3b_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r3b_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH", d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
        for k, v in r["all_conv_kernels"].items():
            if "planes" in k or "pgemm" in k:
                print("   ", k[:150], v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
PY

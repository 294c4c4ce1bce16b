#!/bin/bash
# round 2, call A: full GPU test-suite + parity report (blocked vs single-level accumulation) + Winograd microbench A/B + bench A/B
mkdir -p gpurun_out; rm -f gpurun_out/bench_shape_parity.txt
export PYTHONDONTWRITEBYTECODE=1
rocm-smi --showproductname 2>/dev/null | head -5 > gpurun_out/gpu.txt
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
tail -30 gpurun_out/tests_gpu.log
timeout 600 python tools/parity_report.py > gpurun_out/parity_report.log 2>&1; grep PARITY gpurun_out/parity_report.log
timeout 300 python tools/run_with_lib.py tools/libdbg_NOBLK.so tools/parity_report.py cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 cc_ic_r64 > gpurun_out/parity_report_noblk.log 2>&1; grep PARITY gpurun_out/parity_report_noblk.log | sed 's/^/NOBLK /'
timeout 300 python tools/wino_bench.py > gpurun_out/wino_bench_blk.log 2>&1; cat gpurun_out/wino_bench_blk.log | grep -v amdgpu
ICG_LIB=$PWD/tools/libdbg_NOBLK.so timeout 300 python tools/wino_bench.py > gpurun_out/wino_bench_noblk.log 2>&1; grep -v amdgpu gpurun_out/wino_bench_noblk.log | sed 's/^/NOBLK /'
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_blk.log 2>&1; python - <<'PY'
import json
for f in ("gpurun_out/bench_blk.log",):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f, d["ms_per_step"], d["value"], {k: (v["executed_tflops"], v["ms_per_step"]) for k, v in d["roofline"]["all_conv_kernels"].items() if "planes_kernel" in k})
PY
timeout 300 python tools/run_with_lib.py tools/libdbg_NOBLK.so bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_noblk.log 2>&1; python - <<'PY'
import json
for f in ("gpurun_out/bench_noblk.log",):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); print(f, d["ms_per_step"], d["value"], {k: (v["executed_tflops"], v["ms_per_step"]) for k, v in d["roofline"]["all_conv_kernels"].items() if "planes_kernel" in k})
PY

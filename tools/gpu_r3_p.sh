#!/bin/bash
# round 3, call P: row-streaming small-M GEMM for StyleGAN2's dense layers; cfg4 fp16 / fp32
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm_batched or f16" > gpurun_out/r3p_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3p_kern.log | tail -15 | cut -c1-400
timeout 900 python -m pytest tests/test_stylegan_conv.py tests/test_stylegan2.py tests/test_host_logic_cpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r3p_sg2.log 2>&1; echo "sg2 tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3p_sg2.log | tail -15 | cut -c1-400
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3p_cfg4_fp16.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3p_cfg4.log 2>&1
python - <<'PY'
import json
for tag in ("cfg4_fp16", "cfg4"):
    for l in open("gpurun_out/r3p_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
PY

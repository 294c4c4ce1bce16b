#!/bin/bash
# round 3, call X: fp16 bias gradient of bias_act on icg_colsum_f16 -- kernel test, StyleGAN2 suites, cfg4 --fp16 A/B on one box
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "colsum_f16" > gpurun_out/r3x_kern.log 2>&1; echo "kernel rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3x_kern.log | tail -5 | cut -c1-300
timeout 600 python -m pytest tests/test_stylegan_ops.py tests/test_stylegan_plugin.py tests/test_stylegan_conv.py tests/test_stylegan2.py tests/test_sg2_snapshot.py -m gpu -q -p no:cacheprovider > gpurun_out/r3x_sg2.log 2>&1; echo "sg2 rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3x_sg2.log | tail -5 | cut -c1-300
ICG_FUSED_BIAS_GRAD=0 timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3x_off.log 2>&1
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3x_on.log 2>&1
python - <<'PY'
import json
for tag in ("off", "on"):
    for l in open("gpurun_out/r3x_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); print("BENCH cfg4 --fp16, fused bias gradient", tag, d["ms_per_step"], d["value"])
PY

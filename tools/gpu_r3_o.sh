#!/bin/bash
# round 3, call O: fp16 weight gradient (csrc/hwgrad.hip, transposing LDS reads) -- kernel tests, StyleGAN2 suites, cfg4 fp16 / fp32 bench,
# kernel trace of the fp16 run
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "f16" > gpurun_out/r3o_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3o_kern.log | tail -15 | cut -c1-400
timeout 900 python -m pytest tests/test_stylegan_conv.py tests/test_stylegan2.py -m gpu -q -p no:cacheprovider > gpurun_out/r3o_sg2.log 2>&1; echo "sg2 tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3o_sg2.log | tail -15 | cut -c1-400
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3o_cfg4_fp16.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3o_cfg4.log 2>&1
python - <<'PY'
import json
for tag in ("cfg4_fp16", "cfg4"):
    for l in open("gpurun_out/r3o_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
            for k, v in sorted(r["all_conv_kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:8]:
                print("   %7.2f ms %5d  %6.1f TF  %s" % (v["ms_per_step"], v["launches_per_step"], v["executed_tflops"], k[:100]))
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -o cfg4 -- python $R/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/r3o_rocprof.log 2>&1
cp /tmp/prof4/cfg4_kernel_stats.csv $R/gpurun_out/r3o_cfg4_fp16_kernel_stats.csv; head -40 /tmp/prof4/cfg4_kernel_stats.csv | cut -c1-150

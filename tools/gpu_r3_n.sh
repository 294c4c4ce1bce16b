#!/bin/bash
# round 3, call N: fp16 kernel tests (fixed helper), K<=3 1x1 kernel through the StyleGAN2 suites, cfg4 fp16 bench + kernel trace,
# ds_read_b64_tr_b16 probe
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python tools/probes/tr_probe.py > gpurun_out/r3n_tr_probe.txt 2>&1; echo "probe rc=$?"
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "f16" > gpurun_out/r3n_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3n_kern.log | tail -15 | cut -c1-400
timeout 900 python -m pytest tests/test_stylegan_conv.py tests/test_stylegan2.py -m gpu -q -p no:cacheprovider > gpurun_out/r3n_sg2.log 2>&1; echo "sg2 tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3n_sg2.log | tail -15 | cut -c1-400
timeout 300 python bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3n_cfg4_fp16.log 2>&1
python - <<'PY'
import json
for tag in ("cfg4_fp16",):
    for l in open("gpurun_out/r3n_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH", tag, d["ms_per_step"], d["value"], r["kernel"], r["achieved"], r["frac"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg4 -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --fp16 --steps 16 --warmup 4 --no-cpu-baseline --no-kernel-timer > /tmp/prof_cfg4.log 2>&1
f=$(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3n_cfg4_fp16_kernel_stats.csv; head -45 "$f" | cut -c1-160

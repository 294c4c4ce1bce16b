#!/bin/bash
# round 2, call B: test-suite after the SyncBN rework + error attribution by route + flush-period ablation + bench line check
mkdir -p gpurun_out; rm -f gpurun_out/bench_shape_parity.txt
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/tests_gpu.log
tail -12 gpurun_out/tests_gpu.log
timeout 400 python tools/parity_report.py --routes cfg2_w96_r128 cfg3_w96_r256 cfg1_icgan_res64 > gpurun_out/parity_routes.log 2>&1; grep PARITY gpurun_out/parity_routes.log
for v in F1 F4; do
  timeout 300 python tools/run_with_lib.py tools/libdbg_$v.so tools/parity_report.py cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 > gpurun_out/parity_report_$v.log 2>&1; grep "PARITY.*wino= 0" gpurun_out/parity_report_$v.log | sed "s/^/$v /"
  for l in G.b3 G.b0 G.b5; do ICG_LIB=$PWD/tools/libdbg_$v.so timeout 200 python tools/wino_bench.py $l 2>&1 | grep -v amdgpu | sed "s/^/$v /"; done
done
for l in G.b3 G.b0 G.b5; do timeout 200 python tools/wino_bench.py $l 2>&1 | grep -v amdgpu | sed "s/^/F2 /"; done
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_cfg3.log 2>&1; tail -c 2500 gpurun_out/bench_cfg3.log | head -c 2500
python - <<'PY'
import json
for l in open("gpurun_out/bench_cfg3.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]; r.pop("all_conv_kernels"); print("\nBENCH", d["ms_per_step"], d["value"], json.dumps(r), json.dumps(d["cpu_baseline"]))
PY

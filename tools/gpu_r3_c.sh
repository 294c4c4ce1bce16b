#!/bin/bash
# round 3, call C: streaming plane GEMM + 16-byte epilogue; variants by environment; parity report under the dense criteria
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "plane_gemm" -p no:cacheprovider > gpurun_out/r3c_plane_tests.log 2>&1; echo "plane tests rc=$?"
tail -3 gpurun_out/r3c_plane_tests.log | cut -c1-300
for v in "STREAM1:" "STREAM0:ICG_PGEMM_STREAM=0" "L1_96:ICG_PGEMM_L1_MAXK=96" "L1_192:ICG_PGEMM_L1_MAXK=192" "RUN24:ICG_PGEMM_RUN_KTILES=24" "RUN96:ICG_PGEMM_RUN_KTILES=96"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python tools/pgemm_bench.py nn_only > gpurun_out/r3c_pgemm_$tag.log 2>&1
  echo "== $tag ($envs)"; grep -E "^[GD]\.|^sum" gpurun_out/r3c_pgemm_$tag.log | awk '{printf "%s %s %s %s | ", $1, $2, $(NF-7), $(NF-5)} END {print ""}' | cut -c1-1200
done
timeout 900 python tools/parity_report.py cfg1_icgan_res64 cfg2_w96_r128 cfg3_w96_r256 cfg3_w96_r256_b16 cfg3_w96_r256_b64 > gpurun_out/r3c_parity_report.log 2>&1; echo "parity report rc=$?"
grep -E "PARITY|DENSE" gpurun_out/r3c_parity_report.log | cut -c1-420
timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3c_bench.log 2>&1; echo "bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r3c_bench.log"):
    if l.startswith("{"):
        d = json.loads(l); r = d["roofline"]
        print("BENCH", d["ms_per_step"], d["value"], "uninstrumented", d["config"].get("uninstrumented_ms_per_step"), r["kernel"], r["achieved"], r["frac"], "stale", r.get("traffic_stale"))
        for k, v in r["all_conv_kernels"].items():
            if ("planes" in k or "pgemm" in k) and not k.startswith("composite"):
                print("   ", k[:150], v["executed_tflops"], v["ms_per_step"], v["launches_per_step"])
PY

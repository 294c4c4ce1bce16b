#!/bin/bash
# round 3, call J: non-temporal hints (output-transform loads of the M planes; DMA of an A operand read by one workgroup): A/B on the
# step, StyleGAN2 real-network tests at the calibrated tolerances
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_stylegan2.py -m gpu -q -p no:cacheprovider > gpurun_out/r3j_sg2.log 2>&1; echo "sg2 tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3j_sg2.log | tail -6 | cut -c1-300
for v in "base:" "wnt:ICG_WINO_NT=1" "pnt:ICG_PGEMM_NT=1" "both:ICG_WINO_NT=1 ICG_PGEMM_NT=1" "base2:"; do
  tag=${v%%:*}; envs=${v#*:}
  env $envs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3j_bench_$tag.log 2>&1
done
python - <<'PY'
import json
for tag in ("base", "wnt", "pnt", "both", "base2"):
    for l in open("gpurun_out/r3j_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            ks = r["all_conv_kernels"]
            st = ks.get("void icg_pgemm_nn_stream_kernel<3, 2>(PgemmSP)") or {}
            comp = [v["ms_per_step"] for k, v in ks.items() if k.startswith("composite: wino4_input_kernel + icg_pgemm_nn_stream")]
            print("BENCH %-6s %8.3f instr %8.3f uninstr | stream<3,2> %6.2f ms %6.1f TF | wino4 TN3 composite %s" % (
                tag, d["ms_per_step"], d["config"]["uninstrumented_ms_per_step"], st.get("ms_per_step", 0), st.get("executed_tflops", 0), comp))
PY

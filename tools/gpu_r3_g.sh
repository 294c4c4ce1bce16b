#!/bin/bash
# round 3, call G: ring depth 4 (DMA three K-tiles ahead) A/B; secondary workloads on the second-generation kernels; MFMA-utilisation pass
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$PWD
ICG_PGEMM_NBUF=4 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "plane_gemm" -p no:cacheprovider > gpurun_out/r3g_plane_tests_nbuf4.log 2>&1; echo "plane tests NBUF=4 rc=$?"
tail -2 gpurun_out/r3g_plane_tests_nbuf4.log | cut -c1-200
timeout 300 python tools/pgemm_bench.py > gpurun_out/r3g_pgemm_nbuf3.log 2>&1
ICG_PGEMM_NBUF=4 timeout 300 python tools/pgemm_bench.py > gpurun_out/r3g_pgemm_nbuf4.log 2>&1
paste -d'\n' gpurun_out/r3g_pgemm_nbuf3.log gpurun_out/r3g_pgemm_nbuf4.log | grep -E "N=  192|N=   96|^sum" | cut -c1-150
ICG_PGEMM_NBUF=4 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3g_bench_nbuf4.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3g_bench_nbuf3.log 2>&1
timeout 300 python bench.py --workload cfg2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3g_bench_cfg2.log 2>&1
timeout 300 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3g_bench_cfg5.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --no-cpu-baseline > gpurun_out/r3g_bench_cfg4.log 2>&1
timeout 300 python bench.py --workload cfg4 --steps 16 --warmup 4 --fp16 --no-cpu-baseline > gpurun_out/r3g_bench_cfg4_fp16.log 2>&1
timeout 300 python bench.py --workload sample --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3g_bench_sample.log 2>&1
python - <<'PY'
import json
for w in ("nbuf4", "nbuf3", "cfg2", "cfg5", "cfg4", "cfg4_fp16", "sample"):
    f = "gpurun_out/r3g_bench_%s.log" % w
    ok = False
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); ok = True
            r = d.get("roofline") or {}
            print("BENCH", w, d["ms_per_step"], d["value"], "uninstr", (d.get("config") or {}).get("uninstrumented_ms_per_step"), json.dumps({a: r.get(a) for a in ("kernel", "achieved", "frac")}))
    if not ok:
        print("BENCH", w, "NO JSON LINE"); print(open(f).read()[-1200:])
PY
cd $R && bash tools/gpu_pmc_sq.sh 2>&1 | head -16

#!/bin/bash
# round 3, call T: attention scores + softmax in one kernel (csrc/attn.hip) -- kernel tests, parity suite, A/B bench on one box
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "attn" > gpurun_out/r3t_kern.log 2>&1; echo "kernel tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed|mismatch" gpurun_out/r3t_kern.log | tail -15 | cut -c1-400
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fused-attention > gpurun_out/r3t_bench_off.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3t_bench_on.log 2>&1
python - <<'PY'
import json
for tag in ("off", "on"):
    for l in open("gpurun_out/r3t_bench_%s.log" % tag):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]
            print("BENCH fused attention", tag, d["ms_per_step"], d["value"], "uninstr", d["config"]["uninstrumented_ms_per_step"], r["kernel"], r["achieved"], r["frac"])
PY
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider -k "bench_config or real_widths" > gpurun_out/r3t_parity.log 2>&1; echo "parity rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r3t_parity.log | tail -12 | cut -c1-300
cat > /tmp/attn_time.py <<'PY'
import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import ic_gan_amd._lib as L
def t(fn, it=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / it
for name, B, n, m, d in (("G attention 64x64, ch 384 (d 48), B 64", 64, 4096, 1024, 48), ("D attention 64x64, ch 192 (d 24), B 128", 128, 4096, 1024, 24), ("D attention, B 64", 64, 4096, 1024, 24)):
    th = torch.randn(B, n, d, device="cuda"); ph = torch.randn(B, m, d, device="cuda") * 0.7
    s = torch.empty(B, n, m, device="cuda"); be = torch.empty_like(s)
    def two():
        L.call("icg_gemm_batched", th, ph, s, n, m, d, 0, 1, n * d, m * d, n * m, B, 1.0); L.call("icg_softmax_fwd", s, be, B * n, m)
    def one():
        L.call("icg_attn_scores_softmax", th, ph, be, B, n, m, d)
    a, b = t(two), t(one)
    gb = B * n * m * 4 / 1e9
    print("%-44s scores GEMM + softmax %.3f ms | fused %.3f ms (%.2fx; beta = %.2f GB, %.0f GB/s of beta writes)" % (name, a, b, a / b, gb, gb / b * 1e3))
PY
timeout 300 python /tmp/attn_time.py > gpurun_out/r3t_attn_microbench.txt 2>&1; cat gpurun_out/r3t_attn_microbench.txt | grep -v amdgpu.ids

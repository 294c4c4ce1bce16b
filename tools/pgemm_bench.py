#!/usr/bin/env python
"""Plane-GEMM microbenchmark (tools only): the batched GEMMs over Winograd planes of the cfg3 step, one line per shape, through
icg_plane_gemm -- i.e. on whichever generation the library dispatches (ICG_PGEMM=0 in the environment: first-generation kernels;
tools/gpu_r3_a.sh runs both and prints them side by side).  Also checks each result against fp64 on a row sample.

    python tools/pgemm_bench.py [substring]      ->  name, ms, executed TFLOP/s, fraction of the fp32 MFMA peak, rel. error
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                         # noqa: E402
import ic_gan_amd._lib as L          # noqa: E402
from tools.conv_bench import ev_time  # noqa: E402

PEAK = 157.3
# (name, M = tiles, N = Cout, K = Cin, planes): forward / data-gradient plane GEMMs of the cfg3 step (B = 64 in G, 128 / 64 in D)
SHAPES = [
    ("G.b0.conv2 1536->1536 @8   F4  B64", 64 * 2 * 2, 1536, 1536, 36),
    ("G.b1.conv1 1536->768 up@16 RS  B64", 64 * 4 * 4, 768, 1536, 25),
    ("G.b1.conv2 768->768 @16    F4  B64", 64 * 4 * 4, 768, 768, 36),
    ("G.b2.conv2 768->768 @32    F4  B64", 64 * 8 * 8, 768, 768, 36),
    ("G.b3.conv1 768->384 up@64  RS  B64", 64 * 16 * 16, 384, 768, 25),
    ("G.b3.conv2 384->384 @64    F4  B64", 64 * 16 * 16, 384, 384, 36),
    ("D.b3.conv1 384->768 @32    F4 B128", 128 * 8 * 8, 768, 384, 36),
    ("D.b4.conv1 768->1536 @16   F4 B128", 128 * 4 * 4, 1536, 768, 36),
    ("D.b2.conv1 192->384 @64    F4 B128", 128 * 16 * 16, 384, 192, 36),
    ("G.b4.conv1 384->192 up@128 RS  B64", 64 * 32 * 32, 192, 384, 25),
    ("G.b4.conv2 192->192 @128   F4  B64", 64 * 32 * 32, 192, 192, 36),
    ("D.b1.conv1 96->192 @128    F4 B128", 128 * 32 * 32, 192, 96, 36),
    ("G.b5.conv1 192->96 up@256  RS  B64", 64 * 64 * 64, 96, 192, 25),
    ("G.b5.conv2 96->96 @256     F4  B64", 64 * 64 * 64, 96, 96, 36),
]
NN_ONLY = len(sys.argv) > 1 and sys.argv[1] == "nn_only"
if NN_ONLY:
    sys.argv.pop(1)
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if sys.argv[1] in s[0]]
gen = "first-generation (ICG_PGEMM=0)" if os.environ.get("ICG_PGEMM", "1")[:1] == "0" else "second-generation where it has a tile"
print("plane GEMMs, %s" % gen, flush=True)
tot = 0.0
for name, M, N, K, planes in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    A = torch.randn(planes, M, K, device="cuda", generator=g) / K ** 0.5
    Bm = torch.randn(planes, N, K, device="cuda", generator=g)
    C = torch.empty(planes, M, N, device="cuda")
    t = ev_time(lambda: L.call("icg_plane_gemm", A, Bm, C, M, N, K, planes, 1.0))
    tf = 2.0 * planes * M * N * K / t / 1e12
    rows = torch.arange(0, M, max(M // 61, 1), device="cuda")
    ref = torch.bmm(A[:, rows].double(), Bm.double().transpose(1, 2))
    err = float((C[:, rows].double() - ref).norm() / ref.norm())
    tot += t
    print(f"{name:38s} M={M:7d} N={N:5d} K={K:5d} z={planes:2d}  {t * 1e3:8.3f} ms  {tf:6.1f} TF  {tf / PEAK:5.3f}  err {err:.2e}", flush=True)
print(f"sum {tot * 1e3:.3f} ms")
if NN_ONLY:
    sys.exit(0)

# weight-gradient plane GEMMs: C[z] = A[z]^T B[z], A [K][M] = V planes (tiles x Cin), B [K][N] = transformed dy (tiles x Cout)
TN_SHAPES = [
    ("G.b0.conv2 1536->1536 @8   F4  B64", 1536, 1536, 64 * 2 * 2, 36),
    ("G.b2.conv2 768->768 @32    F4  B64", 768, 768, 64 * 8 * 8, 36),
    ("G.b3.conv1 768->384 up@64  RS  B64", 768, 384, 64 * 16 * 16, 25),
    ("G.b3.conv2 384->384 @64    F4  B64", 384, 384, 64 * 16 * 16, 36),
    ("D.b3.conv1 384->768 @32    F4 B128", 384, 768, 128 * 8 * 8, 36),
    ("D.b2.conv1 192->384 @64    F4 B128", 192, 384, 128 * 16 * 16, 36),
    ("G.b4.conv2 192->192 @128   F4  B64", 192, 192, 64 * 32 * 32, 36),
    ("D.b1.conv1 96->192 @128    F4 B128", 96, 192, 128 * 32 * 32, 36),
    ("G.b5.conv1 192->96 up@256  RS  B64", 192, 96, 64 * 64 * 64, 25),
    ("G.b5.conv2 96->96 @256     F4  B64", 96, 96, 64 * 64 * 64, 36),
]
if len(sys.argv) > 1:
    TN_SHAPES = [s for s in TN_SHAPES if sys.argv[1] in s[0]]
print("---- weight-gradient plane GEMMs (A^T B, split-K + reduction included)")
tot = 0.0
for name, M, N, K, planes in TN_SHAPES:
    g = torch.Generator(device="cuda").manual_seed(2)
    A = torch.randn(planes, K, M, device="cuda", generator=g)
    Bm = torch.randn(planes, K, N, device="cuda", generator=g) / K ** 0.5
    C = torch.empty(planes, M, N, device="cuda")
    nb = L.query("icg_plane_gemm_tn_workspace_bytes", M, N, K, planes)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    t = ev_time(lambda: L.call("icg_plane_gemm_tn", A, Bm, C, M, N, K, planes, ws, nb))
    tf = 2.0 * planes * M * N * K / t / 1e12
    zs = [0, planes - 1]
    ref = torch.bmm(A[zs].double().transpose(1, 2), Bm[zs].double())
    err = float((C[zs].double() - ref).norm() / ref.norm())
    tot += t
    print(f"{name:38s} M={M:5d} N={N:5d} K={K:7d} z={planes:2d}  {t * 1e3:8.3f} ms  {tf:6.1f} TF  {tf / PEAK:5.3f}  err {err:.2e}", flush=True)
print(f"sum {tot * 1e3:.3f} ms")

#!/usr/bin/env python
"""Fused F(4x4,3x3) kernel (csrc/fwino.hip) against the three-kernel composite it replaces, at the narrow layers of cfg3 and
their full batch (tools only).  Per layer and direction: ms of the composite (ICG_FWINO=0), ms of the fused route without and
with the V by-product (ICG_WINO_KEEP_V), executed TFLOP/s of the fused route, rel. L2 of fused vs composite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
from tools.conv_bench import ev_time

KEEP_V = 32
# name, kind, B, H, W (Winograd domain), Cin, Cout (of the GEMM as called), flags
LAYERS = [
    ("G.b5.conv2 96->96 @256 fwd", "plain", 64, 256, 256, 96, 96, 3),
    ("G.b5.conv2 96->96 @256 dgrad", "plain", 64, 256, 256, 96, 96, 0),
    ("G.b4.conv2 192->192 @128 fwd", "plain", 64, 128, 128, 192, 192, 3),
    ("D.b1.conv1 96->192 @128 B128 fwd", "plain", 128, 128, 128, 96, 192, 1),
    ("D.b1.conv1 192->96 @128 B128 dgrad+mask", "plain_mask", 128, 128, 128, 192, 96, 0),
    ("G.b5.conv1 192->96 up@256 fwd", "up", 64, 256, 256, 192, 96, 3),
    ("G.b5.conv1 96->192 pool@256 dgrad", "pool", 64, 256, 256, 96, 192, 0),
    ("G.b4.conv1 192->384.. n/a", None, 0, 0, 0, 0, 0, 0),
    ("D.b1.conv2 192->192 pool@128 B128 fwd", "pool_res", 128, 128, 128, 192, 192, 1),
    ("D.b1.conv2 192->192 up@128 B128 dgrad+mask", "up_mask", 128, 128, 128, 192, 192, 0),
    ("D.b0.conv2 96->96 pool@256 B128 fwd", "pool_res", 128, 256, 256, 96, 96, 1),
]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if sys.argv[1] in l[0]]
dev = "cuda"
for name, kind, B, H, W, Cin, Cout, fl in LAYERS:
    if kind is None:
        continue
    planes = 36 if kind.startswith("plain") else 25
    up = kind.startswith("up")
    pool = kind.startswith("pool")
    Hx, Wx = (H // 2, W // 2) if up else (H, W)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    x = torch.randn(B, Cin, Hx, Wx, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    U = torch.empty(planes, Cout, Cin, device=dev)
    L.call("icg_wino4_weight_transform" if planes == 36 else "icg_wino4r_weight_transform", w, U, Cout, Cin)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1
    bias = torch.randn(Cout, device=dev)
    res = torch.randn(B, Cout, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last) if ("mask" in kind or "res" in kind) else None
    outs = {}
    nb = L.query("icg_conv2d_wino4_workspace_bytes" if planes == 36 else "icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def run(out, keep):
        k = KEEP_V if keep else 0
        if kind == "plain":
            L.call("icg_conv2d_wino4_fprop", x, U, bias, None, out, sc, sh, Cin, B, H, W, Cin, Cout, fl | k, 1.0, ws, nb)
        elif kind == "plain_mask":
            L.call("icg_conv2d_wino4_fprop", x, U, None, res, out, None, None, 0, B, H, W, Cin, Cout, 16, 1.0, ws, nb)
        elif kind == "up":
            L.call("icg_conv2d_up_wino_fprop", x, U, bias, out, sc, sh, Cin, B, H // 2, W // 2, Cin, Cout, fl | k, ws, nb)
        elif kind == "up_mask":       # data gradient of a pool-fused layer: dy at the pooled resolution, ReLU mask in the epilogue
            L.call("icg_conv2d_down_wino_dgrad_relu", x, U, res, out, B, H // 2, W // 2, Cout, Cin, ws, nb)
        elif kind == "pool":          # data gradient of an upsample-fused layer
            L.call("icg_conv2d_up_wino_dgrad", x, U, out, B, H // 2, W // 2, Cout, Cin, ws, nb)
        elif kind == "pool_res":
            L.call("icg_conv2d_down_wino_fprop", x, U, bias, res, out, B, H // 2, W // 2, Cin, Cout, (fl & 1) | k, ws, nb)

    t = {}
    for tag, env, keep in (("composite", "0", False), ("fused", "1", False), ("fused+V", "1", True)):
        os.environ["ICG_FWINO"] = env
        if keep and kind not in ("plain", "up", "pool_res"):
            continue
        out = torch.empty(B, Cout, Ho, Wo, device=dev).contiguous(memory_format=torch.channels_last)
        t[tag] = ev_time(lambda: run(out, keep))
        outs[tag] = out
    flops = 2.0 * planes * B * (H // 4) * (W // 4) * Cin * Cout
    err = float((outs["fused"] - outs["composite"]).norm() / outs["composite"].norm())
    applies = L.lib().icg_fwino_applies(B, H, W, Cin, Cout)
    print(f"{name:44s} applies {applies}  composite {t['composite']*1e3:7.3f} ms  fused {t['fused']*1e3:7.3f} ms ({t['composite']/t['fused']:4.2f}x,"
          f" {flops/t['fused']/1e12:6.1f} TF)" + (f"  fused+V {t['fused+V']*1e3:7.3f} ms" if "fused+V" in t else "") + f"  rel L2 {err:.2e}", flush=True)
    del x, ws, outs
    torch.cuda.empty_cache()

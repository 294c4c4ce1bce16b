#!/usr/bin/env python
"""Winograd vs direct 3x3 convolution on the wide cfg3 layers (tools only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
from tools.conv_bench import ev_time

LAYERS = [("G.b3.conv2 384->384 @64 B64", 64, 64, 64, 384, 384), ("G.b2.conv2 768->768 @32 B64", 64, 32, 32, 768, 768),
          ("D.b3.conv1 384->768 @32 B128", 128, 32, 32, 384, 768), ("D.b4.conv1 768->768 @16 B128", 128, 16, 16, 768, 768),
          ("G.b0.conv2 1536->1536 @8 B64", 64, 8, 8, 1536, 1536), ("G.b4.conv2 192->192 @128 B64", 64, 128, 128, 192, 192),
          ("G.b5.conv2 96->96 @256 B64", 64, 256, 256, 96, 96), ("D.b1.conv1 96->192 @128 B128", 128, 128, 128, 96, 192),
          ("SG2 128->128 @128 B16", 16, 128, 128, 128, 128), ("SG2 256->256 @64 B16", 16, 64, 64, 256, 256),
          ("SG2 512->512 @32 B16", 16, 32, 32, 512, 512), ("SG2 512->512 @16 B16", 16, 16, 16, 512, 512),
          ("SG2 512->512 @8 B16", 16, 8, 8, 512, 512), ("SG2 128->128 @128 B8", 8, 128, 128, 128, 128)]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if sys.argv[1] in l[0]]
for name, B, H, W, Cin, Cout in LAYERS:
    dev = "cuda"
    x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
    out = torch.empty(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    out2 = torch.empty_like(out)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1
    fl = int(os.environ.get("WINO_BENCH_FLAGS", "3"))      # 3: BN affine + ReLU prologue (G);  1: ReLU only (D: second-generation implicit GEMM)
    nf = L.query("icg_conv2d_fprop_workspace_bytes", B, H, W, Cin, Cout, 3, fl)
    wf = torch.empty(max(nf, 16), dtype=torch.uint8, device=dev)
    t_d = ev_time(lambda: L.call("icg_conv2d_fprop_ws", x, w, None, None, out, sc, sh, Cin, B, H, W, Cin, Cout, 3, fl, 1.0, wf, nf))
    U = torch.empty(16, Cout, Cin, device=dev)
    L.call("icg_wino_weight_transform", w, U, Cout, Cin)
    nb = L.query("icg_conv2d_wino_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    t_w = ev_time(lambda: L.call("icg_conv2d_wino_fprop", x, U, None, None, out2, sc, sh, Cin, B, H, W, Cin, Cout, fl, 1.0, ws, nb))
    err = float((out2 - out).norm() / out.norm())
    U4 = torch.empty(36, Cout, Cin, device=dev)
    L.call("icg_wino4_weight_transform", w, U4, Cout, Cin)
    nb4 = L.query("icg_conv2d_wino4_workspace_bytes", B, H, W, Cin, Cout)
    ws4 = torch.empty(nb4, dtype=torch.uint8, device=dev)
    out3 = torch.empty_like(out)
    t_4 = ev_time(lambda: L.call("icg_conv2d_wino4_fprop", x, U4, None, None, out3, sc, sh, Cin, B, H, W, Cin, Cout, fl, 1.0, ws4, nb4))
    err4 = float((out3 - out).norm() / out.norm())
    print(f"{name:34s} direct {t_d*1e3:7.3f} ms  F(2,3) {t_w*1e3:7.3f} ms ({t_d/t_w:4.2f}x, {err:.1e})  F(4,3) {t_4*1e3:7.3f} ms ({t_d/t_4:4.2f}x, {err4:.1e})", flush=True)
print("---- weight gradient")
for name, B, H, W, Cin, Cout in LAYERS:
    dev = "cuda"
    x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1
    dw1, dw2 = torch.empty(9 * Cin * Cout, device=dev), torch.empty(9 * Cin * Cout, device=dev)
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, 3)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    t_d = ev_time(lambda: L.call("icg_conv2d_wgrad", x, dy, dw1, sc, sh, Cin, B, H, W, Cin, Cout, 3, 3, ws, nb))
    nbw = L.query("icg_conv2d_wino_wgrad_workspace_bytes", B, H, W, Cin, Cout)
    wsw = torch.empty(nbw, dtype=torch.uint8, device=dev)
    t_w = ev_time(lambda: L.call("icg_conv2d_wino_wgrad", x, dy, dw2, sc, sh, Cin, B, H, W, Cin, Cout, 3, wsw, nbw))
    err = float((dw2 - dw1).norm() / dw1.norm())
    dw3 = torch.empty(9 * Cin * Cout, device=dev)
    nb4 = L.query("icg_conv2d_wino4_wgrad_workspace_bytes", B, H, W, Cin, Cout)
    ws4 = torch.empty(nb4, dtype=torch.uint8, device=dev)
    t_4 = ev_time(lambda: L.call("icg_conv2d_wino4_wgrad", x, dy, dw3, sc, sh, Cin, B, H, W, Cin, Cout, 3, ws4, nb4))
    err4 = float((dw3 - dw1).norm() / dw1.norm())
    print(f"{name:34s} wgrad direct {t_d*1e3:7.3f} ms  F(2,3) {t_w*1e3:7.3f} ms ({t_d/t_w:4.2f}x, {err:.1e})  F(4,3) {t_4*1e3:7.3f} ms ({t_d/t_4:4.2f}x, {err4:.1e})  ws {nb4>>20} MiB", flush=True)

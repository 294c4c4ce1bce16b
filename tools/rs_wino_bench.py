#!/usr/bin/env python
"""Resample-fused 3x3 layers of cfg3: 2x2-phase / 4x4-stride-2 forms vs the 25-plane F(4x4,3x3) Winograd form (tools only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
from tools.conv_bench import ev_time

dev = "cuda"
UP = [("G.b0.conv1 1536->1536 4->8", 64, 4, 1536, 1536), ("G.b1.conv1 1536->768 8->16", 64, 8, 1536, 768),
      ("G.b2.conv1 768->768 16->32", 64, 16, 768, 768), ("G.b3.conv1 768->384 32->64", 64, 32, 768, 384),
      ("G.b4.conv1 384->192 64->128", 64, 64, 384, 192), ("G.b5.conv1 192->96 128->256", 64, 128, 192, 96)]
DOWN = [("D.b0.conv2 96->96 256->128", 128, 128, 96, 96), ("D.b1.conv2 192->192 128->64", 128, 64, 192, 192),
        ("D.b2.conv2 384->384 64->32", 128, 32, 384, 384), ("D.b3.conv2 768->768 32->16", 128, 16, 768, 768),
        ("D.b4.conv2 1536->1536 16->8", 128, 8, 1536, 1536)]
sel = sys.argv[1] if len(sys.argv) > 1 else ""


def buf(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=dev)


for name, B, Hs, Cin, Cout in UP:
    if sel not in name:
        continue
    x = torch.randn(B, Cin, Hs, Hs, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, 2 * Hs, 2 * Hs, device=dev).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(dy)
    da = torch.empty_like(x)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1
    wp = torch.randn(16 * Cout * Cin, device=dev) * 0.02
    U = torch.randn(25 * Cout * Cin, device=dev) * 0.02
    dwp, dw = torch.empty(16 * Cin * Cout, device=dev), torch.empty(9 * Cin * Cout, device=dev)
    nbp = L.query("icg_conv2d_up_wgrad_workspace_bytes", B, Hs, Hs, Cin, Cout)
    wsp = buf(nbp)
    nf = L.query("icg_conv2d_rs_wino_workspace_bytes", B, 2 * Hs, 2 * Hs, Cin, Cout)
    nd = L.query("icg_conv2d_rs_wino_workspace_bytes", B, 2 * Hs, 2 * Hs, Cout, Cin)
    nw = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, 2 * Hs, 2 * Hs, Cin, Cout)
    ws = buf(max(nf, nd, nw))
    t = [ev_time(lambda: L.call("icg_conv2d_up_fprop", x, wp, None, out, sc, sh, Cin, B, Hs, Hs, Cin, Cout, 3)),
         ev_time(lambda: L.call("icg_conv2d_up_wino_fprop", x, U, None, out, sc, sh, Cin, B, Hs, Hs, Cin, Cout, 3, ws, nf)),
         ev_time(lambda: L.call("icg_conv2d_up_dgrad", dy, wp, da, B, Hs, Hs, Cin, Cout)),
         ev_time(lambda: L.call("icg_conv2d_up_wino_dgrad", dy, U, da, B, Hs, Hs, Cin, Cout, ws, nd)),
         ev_time(lambda: L.call("icg_conv2d_up_wgrad", x, dy, dwp, sc, sh, Cin, B, Hs, Hs, Cin, Cout, 3, wsp, nbp)),
         ev_time(lambda: L.call("icg_conv2d_up_wino_wgrad", x, dy, dw, sc, sh, Cin, B, Hs, Hs, Cin, Cout, 3, ws, nw))]
    print(f"{name:30s} B{B:<4d} fprop phase {t[0]*1e3:6.3f} wino {t[1]*1e3:6.3f} ({t[0]/t[1]:4.2f}x) | dgrad {t[2]*1e3:6.3f} / {t[3]*1e3:6.3f} "
          f"({t[2]/t[3]:4.2f}x) | wgrad {t[4]*1e3:6.3f} / {t[5]*1e3:6.3f} ({t[4]/t[5]:4.2f}x)", flush=True)
    del x, dy, out, da, ws, wsp

for name, B, Hp, Cin, Cout in DOWN:
    if sel not in name:
        continue
    x = torch.randn(B, Cin, 2 * Hp, 2 * Hp, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, Cout, Hp, Hp, device=dev).contiguous(memory_format=torch.channels_last)
    out = torch.empty_like(dy)
    res = torch.randn_like(dy)
    da = torch.empty_like(x)
    wp = torch.randn(16 * Cout * Cin, device=dev) * 0.02
    U = torch.randn(25 * Cout * Cin, device=dev) * 0.02
    dwp, dw = torch.empty(16 * Cin * Cout, device=dev), torch.empty(9 * Cin * Cout, device=dev)
    nbp = L.query("icg_conv2d_down_wgrad_workspace_bytes", B, Hp, Hp, Cin, Cout)
    wsp = buf(nbp)
    nf = L.query("icg_conv2d_rs_wino_workspace_bytes", B, 2 * Hp, 2 * Hp, Cin, Cout)
    nw = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, 2 * Hp, 2 * Hp, Cin, Cout)
    ws = buf(max(nf, nw))
    t = [ev_time(lambda: L.call("icg_conv2d_down_fprop", x, wp, None, res, out, B, Hp, Hp, Cin, Cout, 1)),
         ev_time(lambda: L.call("icg_conv2d_down_wino_fprop", x, U, None, res, out, B, Hp, Hp, Cin, Cout, 1, ws, nf)),
         ev_time(lambda: L.call("icg_conv2d_down_dgrad", dy, wp, da, B, Hp, Hp, Cin, Cout)),
         ev_time(lambda: L.call("icg_conv2d_down_wino_dgrad", dy, U, da, B, Hp, Hp, Cin, Cout, ws, nf)),
         ev_time(lambda: L.call("icg_conv2d_down_wgrad", x, dy, dwp, B, Hp, Hp, Cin, Cout, 1, wsp, nbp)),
         ev_time(lambda: L.call("icg_conv2d_down_wino_wgrad", x, dy, dw, B, Hp, Hp, Cin, Cout, 1, ws, nw))]
    print(f"{name:30s} B{B:<4d} fprop 4x4s2 {t[0]*1e3:6.3f} wino {t[1]*1e3:6.3f} ({t[0]/t[1]:4.2f}x) | dgrad {t[2]*1e3:6.3f} / {t[3]*1e3:6.3f} "
          f"({t[2]/t[3]:4.2f}x) | wgrad {t[4]*1e3:6.3f} / {t[5]*1e3:6.3f} ({t[4]/t[5]:4.2f}x)", flush=True)
    del x, dy, out, da, ws, wsp

#!/usr/bin/env python
"""Barrier timeline of workgroup 0 of the fused Winograd kernel (tools only; needs tools/libdbg_FWT.so = build_dbg.sh FWT):
shader-clock timestamps of a consumer wave (wave 0) and a producer wave (wave 8) at every barrier arrival / departure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
L.LIB_PATH = os.path.join(ROOT, "tools", "libdbg_FWT.so")
B, H, W, Cin, Cout = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 256, 256, 96, 96)))
dev = "cuda"
x = torch.randn(B, Cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(Cout, 3, 3, Cin, device=dev) / (9 * Cin) ** 0.5
U = torch.empty(36, Cout, Cin, device=dev)
L.call("icg_wino4_weight_transform", w, U, Cout, Cin)
Uf = torch.empty_like(U)
L.call("icg_fwino_pack_weights", U, Uf, 36, Cin, Cout)
sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev) * 0.1
out = torch.empty(B, Cout, H, W, device=dev).contiguous(memory_format=torch.channels_last)
tr = torch.zeros(512, dtype=torch.int64, device=dev)
for it in range(3):
    tr.zero_()
    os.environ["ICG_FWINO_TRACE_PTR"] = str(tr.data_ptr())
    L.call("icg_fwino_conv", x, Uf, None, None, out, sc, sh, Cin, B, H, W, Cin, Cout, 3, 1.0, 0, 0, None)
    torch.cuda.synchronize()
t = tr.cpu().tolist()
for role, base in (("consumer wave 0", 0), ("producer wave 8", 256)):
    v = [a for a in t[base:base + 250] if a != 0]
    t0 = v[0] & ((1 << 63) - 1)
    print(role, "(clocks since its first record; A = barrier arrival, D = departure, M = transform done)")
    ev, k = [], 0
    for a in v:
        if a < 0 or a >> 63:
            ev.append(("M", (a & ((1 << 63) - 1)) - t0))
        else:
            ev.append(("A" if k % 2 == 0 else "D", a - t0)); k += 1
    print("  " + "  ".join("%s%d" % e for e in ev[:120]))

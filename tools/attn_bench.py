#!/usr/bin/env python
"""Attention block kernels at the cfg3 shapes (tools only): scores + softmax and the backward dS = softmax'(dO V^T) in their fused
forms (csrc/attn.hip) against the GEMM + stand-alone softmax forms they replace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ic_gan_amd._lib as L
from tools.conv_bench import ev_time

dev = "cuda"
for name, B, n, m, d, dv in (("G attention (384 ch @64x64, B 64)", 64, 4096, 1024, 48, 192), ("D attention (192 ch @64x64, B 128)", 128, 4096, 1024, 24, 96),
                             ("D attention, G step (B 64)", 64, 4096, 1024, 24, 96)):
    theta = torch.randn(B, n, d, device=dev); phi = torch.randn(B, m, d, device=dev) * 0.7
    g = torch.randn(B, m, dv, device=dev) * 0.5; do = torch.randn(B, n, dv, device=dev)
    beta = torch.empty(B, n, m, device=dev); s = torch.empty_like(beta)
    t_f = ev_time(lambda: L.call("icg_attn_scores_softmax", theta, phi, beta, B, n, m, d))
    def fwd_old():
        L.call("icg_gemm_batched", theta, phi, s, n, m, d, 0, 1, n * d, m * d, n * m, B, 1.0)
        L.call("icg_softmax_fwd", s, beta, B * n, m)
    t_fo = ev_time(fwd_old)
    ds = torch.empty_like(beta); ds2 = torch.empty_like(beta)
    def bwd_old():
        L.call("icg_gemm_batched", do, g, s, n, m, dv, 0, 1, n * dv, m * dv, n * m, B, 1.0)
        L.call("icg_softmax_bwd", beta, s, ds, B * n, m)
    t_bo = ev_time(bwd_old)
    t_b = ev_time(lambda: L.call("icg_attn_dscores", do, g, beta, ds2, B, n, m, dv))
    err = float((ds2 - ds).norm() / ds.norm())
    fl = 2.0 * B * n * m * dv
    print(f"{name:38s} scores+softmax {t_fo*1e3:6.3f} -> {t_f*1e3:6.3f} ms | dS: GEMM + softmax_bwd {t_bo*1e3:6.3f} -> fused {t_b*1e3:6.3f} ms "
          f"({fl/t_b/1e12:5.1f} TF, {(8.0*B*n*m)/t_b/1e9:5.0f} GB/s of beta + dS)  rel L2 {err:.1e}", flush=True)
    del beta, s, ds, ds2
    torch.cuda.empty_cache()

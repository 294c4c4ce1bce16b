#!/usr/bin/env python
"""Goldens for SURVEY §8a row a19 from the UNMODIFIED reference running on CPU:
  torch_utils/ops/conv2d_gradfix.py   conv2d / conv_transpose2d        (falls back to torch.nn.functional on CPU)
  torch_utils/ops/conv2d_resample.py  conv2d_resample                  (upfirdn2d -> its _upfirdn2d_ref)
  training/networks.py                modulated_conv2d                 (both fused_modconv settings)
Outputs, first-order gradients, and second-order gradients (gradient of a scalar built from the first-order ones — what
R1 and path-length regularisation do).  Build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stylegan_conv.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/stylegan2_ada_pytorch")
import numpy as np
import torch
from torch_utils.ops import conv2d_gradfix as ref_cg, conv2d_resample as ref_cr, upfirdn2d as ref_up
from training import networks as ref_net

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.stylegan_cases import CONV, MODCONV, RESAMPLE, rnd   # noqa: E402


def second_order(out, key, y, inputs, names, seed):
    """y -> first-order grads wrt inputs with a random cotangent (kept in graph) -> scalar -> second-order grads."""
    dy = rnd(tuple(y.shape), seed).requires_grad_(True)
    g1 = torch.autograd.grad(y, inputs, dy, create_graph=True)
    for n, g in zip(names, g1):
        out[f"{key}d{n}"] = g.detach().numpy()
    scalar = sum((g * rnd(tuple(g.shape), seed + 1 + i)).sum() for i, g in enumerate(g1))
    g2 = torch.autograd.grad(scalar, list(inputs) + [dy], allow_unused=True)
    for n, g, ref in zip(list(names) + ["y"], g2, list(inputs) + [dy]):
        out[f"{key}dd{n}"] = (g if g is not None else torch.zeros_like(ref)).detach().numpy()


def main():
    out = {}
    for i, (n, ci, h, w, co, r, s, p, tr, op) in enumerate(CONV):
        x = rnd((n, ci, h, w), 100 + i).requires_grad_(True)
        wt = (rnd((ci, co, r, r) if tr else (co, ci, r, r), 200 + i) * (ci * r * r) ** -0.5).requires_grad_(True)
        if tr:
            y = ref_cg.conv_transpose2d(x, wt, stride=s, padding=p, output_padding=op)
        else:
            y = ref_cg.conv2d(x, wt, stride=s, padding=p)
        out[f"conv/{i}/y"] = y.detach().numpy()
        second_order(out, f"conv/{i}/", y, (x, wt), ("x", "w"), 300 + 10 * i)

    for i, (n, ci, h, w, co, k, up, down, pad, fw, ff, taps) in enumerate(RESAMPLE):
        x = rnd((n, ci, h, w), 400 + i).requires_grad_(True)
        wt = (rnd((co, ci, k, k), 500 + i) * (ci * k * k) ** -0.5).requires_grad_(True)
        f = ref_up.setup_filter(taps) if taps is not None else None
        y = ref_cr.conv2d_resample(x, wt, f=f, up=up, down=down, padding=pad, flip_weight=fw, flip_filter=ff)
        out[f"rs/{i}/y"] = y.detach().numpy()
        second_order(out, f"rs/{i}/", y, (x, wt), ("x", "w"), 600 + 10 * i)

    for i, (n, ci, h, w, co, k, up, demod, noise, fused) in enumerate(MODCONV):
        x = rnd((n, ci, h, w), 700 + i).requires_grad_(True)
        wt = rnd((co, ci, k, k), 800 + i).requires_grad_(True)
        st = (rnd((n, ci), 900 + i) * 0.5 + 1.0).requires_grad_(True)
        nz = rnd((n, 1, h * up, w * up), 950 + i) * 0.1 if noise else None
        f = ref_up.setup_filter([1, 3, 3, 1])
        y = ref_net.modulated_conv2d(x=x, weight=wt, styles=st, noise=nz, up=up, padding=k // 2, resample_filter=f,
                                     demodulate=demod, flip_weight=(up == 1), fused_modconv=fused)
        out[f"mc/{i}/y"] = y.detach().numpy()
        second_order(out, f"mc/{i}/", y, (x, wt, st), ("x", "w", "s"), 1000 + 10 * i)

    path = os.path.join(HERE, "stylegan_conv.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")


if __name__ == "__main__":
    main()

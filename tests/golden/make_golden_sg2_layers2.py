#!/usr/bin/env python
"""Goldens for SURVEY 8(f) N1's second-order nodes from the UNMODIFIED reference on CPU, in FLOAT64:
training/networks.py SynthesisLayer (361-444) and ToRGBLayer (449-486) as the synthesis network runs them during the path-length phase
(`fused_modconv=False` when differentiated twice, training/networks.py:560-563; the image accumulation of SynthesisBlock.forward
618-622 for toRGB), differentiated the way training/loss.py:120-139 does:

    y            = layer(x, w)                                       [+ img for toRGB]
    g, gx        = d sum(y * r * rs) / d(w, x)        create_graph=True            (first-order: the path-length vector is g)
    scalar       = |g|^2 / 2 + <g, q> + <gx, qx> + 0.1 <y, qy>
    second order = d scalar / d(x, w, rs, [img], every parameter)

The reference runs in float64 here (its CPU fallbacks are plain torch operators), so the fixture is the mathematical answer to ~1e-15;
it is stored as float32 (6e-8 relative), three orders below the 1e-5 the fp32 HIP path is held to.
Build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sg2_layers2.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/stylegan2_ada_pytorch")
import numpy as np
import torch
from training import networks as ref_net

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import stylegan_cases as SC   # noqa: E402
from tests.stylegan_cases import SG2_LAYER2, sg2_layer2_inputs, sg2_layer2_run as run_case   # noqa: E402


def kink_margin(idx, case, res):
    kind, clamp = case[0], case[7]
    t = res["y"] if kind == "synthesis" else res["y"] - sg2_layer2_inputs(case, idx)[2].double().numpy()
    margin = float(np.abs(t).min()) if kind == "synthesis" else np.inf
    if clamp is not None:
        g = clamp          # (bias_act clamps its result, after the activation gain: the kink of y is at +-conv_clamp)
        inside = np.abs(t) < g
        if inside.any() and (~inside).any():
            margin = min(margin, float(np.abs(np.abs(t[inside]) - g).min()))
    return margin


def ref_layer(case):
    kind, cin, cout, wd, res, up, noise_mode, clamp, n = case
    if kind == "synthesis":
        return ref_net.SynthesisLayer(cin, cout, w_dim=wd, resolution=res, up=up, conv_clamp=clamp)
    return ref_net.ToRGBLayer(cin, cout, w_dim=wd, conv_clamp=clamp)


def main():
    out = {}
    for idx, case in enumerate(SG2_LAYER2):
        res = run_case(idx, case, ref_layer)
        if "--search" in sys.argv:                    # find the salts once; they are then written into tests/stylegan_cases.py by hand
            while kink_margin(idx, case, res) <= 5e-6:
                SC.SG2_LAYER2_SALT[idx] += 1
                res = run_case(idx, case, ref_layer)
            print("salt", idx, SC.SG2_LAYER2_SALT[idx])
        for k, v in res.items():
            out["%d/%s" % (idx, k)] = v.astype(np.float32)
        # conditioning of the comparison: distance of the nearest activation to a kink of the layer's nonlinearity (lrelu at 0, the clamp
        # at +-conv_clamp).  An fp32 implementation that lands one activation on the other side changes that sample's gradients by ~1e-4
        # (profiles/r06_sg2_nondeterminism.txt): the cases are chosen so that no activation is within 5e-6.
        margin = kink_margin(idx, case, res)
        out["%d/kink_margin" % idx] = np.float32(margin)
        assert margin > 5e-6, (idx, margin)
        print(idx, "kink margin %.2e" % margin)
        print(idx, case, {k: "%.3e" % float(np.abs(v).max()) for k, v in res.items() if k in ("y", "g", "dd_w")})
    path = os.path.join(HERE, "sg2_layers_second_order.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")


if __name__ == "__main__":
    main()

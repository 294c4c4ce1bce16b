#!/usr/bin/env python
"""Goldens for the StyleGAN2 custom ops from the UNMODIFIED reference's own reference implementations
(stylegan2_ada_pytorch/torch_utils/ops/bias_act.py::_bias_act_ref, upfirdn2d.py::_upfirdn2d_ref, impl='ref'),
including first- and second-order gradients obtained by autograd through them.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stylegan_ops.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/stylegan2_ada_pytorch")
import numpy as np
import torch
from torch_utils.ops import bias_act as ref_ba, upfirdn2d as ref_up

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.stylegan_cases import ACTS, UPFIR, rnd   # noqa: E402


def main():
    out = {}
    for ai, act in enumerate(ACTS):
        for ci, clamp in enumerate([None, 0.7]):
            x = rnd((3, 6, 5, 5), 10 + ai, 1.5).requires_grad_(True)
            b = rnd((6,), 20 + ai, 0.5).requires_grad_(True)
            dy = rnd((3, 6, 5, 5), 30 + ai).requires_grad_(True)
            d2 = rnd((3, 6, 5, 5), 40 + ai)
            y = ref_ba.bias_act(x, b, act=act, clamp=clamp, impl="ref")
            dx, db = torch.autograd.grad(y, (x, b), dy, create_graph=True)
            ddx, ddy = torch.autograd.grad((dx * d2).sum(), (x, dy), allow_unused=True)
            k = f"ba/{act}/{ci}/"
            out[k + "y"], out[k + "dx"], out[k + "db"] = y.detach().numpy(), dx.detach().numpy(), db.detach().numpy()
            out[k + "ddx"] = (ddx if ddx is not None else torch.zeros_like(x)).numpy()
            out[k + "ddy"] = ddy.detach().numpy()
    for i, (n, c, h, w, taps, up, down, pad, flip, gain) in enumerate(UPFIR):
        x = rnd((n, c, h, w), 50 + i).requires_grad_(True)
        f = ref_up.setup_filter(taps, flip_filter=False)
        y = ref_up.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl="ref")
        dy = rnd(tuple(y.shape), 60 + i)
        (dx,) = torch.autograd.grad(y, x, dy)
        out[f"up/{i}/y"], out[f"up/{i}/dx"], out[f"up/{i}/f"] = y.detach().numpy(), dx.numpy(), f.numpy()
    # convenience wrappers
    x = rnd((2, 3, 8, 8), 70)
    f = ref_up.setup_filter([1, 3, 3, 1])
    out["wrap/upsample2d"] = ref_up.upsample2d(x, f, impl="ref").numpy()
    out["wrap/downsample2d"] = ref_up.downsample2d(x, f, impl="ref").numpy()
    out["wrap/filter2d"] = ref_up.filter2d(x, f, impl="ref").numpy()
    path = os.path.join(HERE, "stylegan_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")
    typed()


def typed():
    """fp16 / fp64 STORAGE goldens for the plugin seam (tests/test_stylegan_plugin.py).  The CUDA plugins compute in fp32 for
    half tensors and round once (bias_act.cu:18-21), which the reference's `impl='ref'` path evaluated on the fp16-rounded
    inputs UPCAST to fp32 reproduces up to that final rounding; fp64 tensors are evaluated in fp64.  Stored unrounded (fp32 /
    fp64): the test applies the storage rounding tolerance."""
    out = {}
    for tag, st, ct in (("f16", torch.float16, torch.float32), ("f64", torch.float64, torch.float64)):
        q = lambda t: t.to(st).to(ct)          # storage rounding, then the plugin's compute type
        for ai, act in enumerate(ACTS):
            for ci, clamp in enumerate([None, 0.7]):
                x = q(rnd((3, 6, 5, 5), 10 + ai, 1.5)).requires_grad_(True)
                b = q(rnd((6,), 20 + ai, 0.5)).requires_grad_(True)
                dy = q(rnd((3, 6, 5, 5), 30 + ai)).requires_grad_(True)
                d2 = q(rnd((3, 6, 5, 5), 40 + ai))
                y = ref_ba.bias_act(x, b, act=act, clamp=clamp, impl="ref")
                dx, db = torch.autograd.grad(y, (x, b), dy, create_graph=True)
                ddx, ddy = torch.autograd.grad((dx * d2).sum(), (x, dy), allow_unused=True)
                k = f"{tag}/ba/{act}/{ci}/"
                out[k + "y"], out[k + "dx"] = y.detach().numpy(), dx.detach().numpy()
                out[k + "ddx"] = (ddx if ddx is not None else torch.zeros_like(x)).numpy()
                out[k + "ddy"] = ddy.detach().numpy()
        for i, (n, c, h, w, taps, up, down, pad, flip, gain) in enumerate(UPFIR):
            x = q(rnd((n, c, h, w), 50 + i)).requires_grad_(True)
            f = ref_up.setup_filter(taps, flip_filter=False)
            y = ref_up.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain, impl="ref")
            dy = q(rnd(tuple(y.shape), 60 + i))
            (dx,) = torch.autograd.grad(y, x, dy)
            out[f"{tag}/up/{i}/y"], out[f"{tag}/up/{i}/dx"] = y.detach().numpy(), dx.numpy()
    path = os.path.join(HERE, "stylegan_ops_typed.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", len(out), "arrays")


if __name__ == "__main__":
    if "--typed-only" in sys.argv:
        typed()
    else:
        main()

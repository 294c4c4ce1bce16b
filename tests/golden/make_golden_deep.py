#!/usr/bin/env python
"""Goldens for BigGAN-deep from the UNMODIFIED reference (BigGAN_PyTorch/BigGANdeep.py) on CPU: state_dict specs, G_D outputs
in training mode, and the parameter gradients of one discriminator step and one generator step with the hinge losses.
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_deep.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (reference import path + stubs, fingerprint packing)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import BigGANdeep as RefDeep  # noqa: E402
import losses as ref_losses  # noqa: E402
from oracle import synth  # noqa: E402

BASE = dict(dim_z=16, shared_dim=8, G_shared=True, hier=True, n_classes=10, SN_eps=1e-6, BN_eps=1e-5, adam_eps=1e-6,
            G_ch=8, D_ch=8, G_depth=2, D_depth=2, skip_init=True)
CASES = {
    "deep_r32": dict(resolution=32, G_attn="16", D_attn="16"),
    "deep_r64": dict(resolution=64, G_attn="32", D_attn="32"),
}
B = 4


def inputs(cfg):
    rs = np.random.RandomState(3)
    z = torch.from_numpy(rs.standard_normal((B, cfg["dim_z"])).astype(np.float32))
    gy = torch.from_numpy(rs.randint(0, cfg["n_classes"], size=B).astype(np.int64))
    x = torch.from_numpy(((rs.randint(0, 256, size=(B, 3, cfg["resolution"], cfg["resolution"])) / 255.0 - 0.5) * 2).astype(np.float32))
    dy = torch.from_numpy(rs.randint(0, cfg["n_classes"], size=B).astype(np.int64))
    return z, gy, x, dy


def run(name, over):
    cfg = dict(BASE, **over)
    G = RefDeep.Generator(**cfg)
    D = RefDeep.Discriminator(**cfg)
    gspec, dspec = synth.spec_of(G.state_dict()), synth.spec_of(D.state_dict())
    G.load_state_dict(synth.synth_state(gspec, seed=11))
    D.load_state_dict(synth.synth_state(dspec, seed=22))
    GD = RefDeep.G_D(G, D)
    G.train(); D.train()
    z, gy, x, dy = inputs(cfg)
    out = {"cfg": json.dumps({k: v for k, v in cfg.items()}), "gspec": json.dumps(gspec), "dspec": json.dumps(dspec)}
    # discriminator step
    D_fake, D_real = GD(z, gy, x, dy, train_G=False)
    l_real, l_fake = ref_losses.loss_hinge_dis(D_fake, D_real)
    (l_real + l_fake).backward()
    out["d_step/D_fake"], out["d_step/D_real"] = D_fake.detach().numpy(), D_real.detach().numpy()
    for k, v in MG.pack({n: p.grad for n, p in D.named_parameters()}).items():
        out["d_step/D_grad/" + k] = v
    for k, v in MG.pack(G.state_dict()).items():
        out["d_step/G_state/" + k] = v
    D.zero_grad()
    # generator step
    D_out, G_z = GD(z, gy, train_G=True, return_G_z=True)
    ref_losses.loss_hinge_gen(D_out).backward()
    out["g_step/G_z"], out["g_step/D_out"] = G_z.detach().numpy(), D_out.detach().numpy()
    for k, v in MG.pack({n: p.grad for n, p in G.named_parameters()}).items():
        out["g_step/G_grad/" + k] = v
    path = os.path.join(HERE, f"biggan_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    for name, over in CASES.items():
        run(name, over)

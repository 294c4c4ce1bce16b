#!/usr/bin/env python
"""Generate tests/golden/biggan_*.npz by running the UNMODIFIED reference
(BigGAN_PyTorch/{BigGAN,layers,losses,train_fns,utils}.py) on CPU.

Runs only in the build container (needs /root/reference); the GPU box uses the
committed .npz files.  Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is recorded per case (SURVEY.md §8c — the reference has no goldens of its own):
  * the (name, shape) list of G / D state_dict  -> checkpoint-layout contract
  * G(z,y,f) in train mode on fresh synthetic weights (full tensor for the small
    cases, fingerprints otherwise), D(G(z)) logits, per-block activation fingerprints
  * N calls of train_fns.GAN_training_function.train(x,y,f): the three losses and
    fingerprints (sum, sum-of-squares, 64 strided samples) of every state_dict entry of
    G, D and G_ema after each step, and of every parameter gradient of step 1.
Weights/inputs come from oracle.synth (numpy RandomState: platform-stable).
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
for m in ["torchvision", "torchvision.transforms", "torchvision.utils"]:
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path[:0] = ["/root/reference", "/root/reference/BigGAN_PyTorch"]

import numpy as np
import torch

import BigGAN_PyTorch.BigGAN as RefBigGAN   # noqa: E402
import train_fns as ref_train_fns            # noqa: E402
import utils as ref_utils                    # noqa: E402
from oracle import synth                     # noqa: E402

BASE = dict(
    dim_z=120, shared_dim=16, shared_dim_feat=32, G_shared=True, G_shared_feat=True, hier=True,
    n_classes=10, SN_eps=1e-6, BN_eps=1e-5, adam_eps=1e-6, G_lr=1e-3, D_lr=2e-3, G_B1=0.0, G_B2=0.999,
    D_B1=0.0, D_B2=0.999, ema_decay=0.9, ema_start=0, ema=True, toggle_grads=True, num_D_steps=1,
    num_D_accumulations=1, num_G_accumulations=1, split_D=False, DiffAugment="", DA=False, D_ortho=0.0,
    G_ortho=0.0, G_ch=8, D_ch=8, G_attn="32", D_attn="32",
)
CASES = {
    # name: (overrides, G batch, steps, store_full_image)
    "cc_ic_r64": (dict(resolution=64, class_cond=True, instance_cond=True), 4, 2, True),
    "ic_r64_acc2": (dict(resolution=64, class_cond=False, instance_cond=True,
                         num_D_accumulations=2, num_G_accumulations=2), 3, 2, True),
    "cc_r32_flat": (dict(resolution=32, class_cond=True, instance_cond=False, hier=False,
                         G_attn="16", D_attn="16", shared_dim=0, dim_z=24), 4, 1, True),
    "cc_ic_r128": (dict(resolution=128, class_cond=True, instance_cond=True, G_attn="64", D_attn="64"),
                   4, 1, False),
    "cc_ic_r256": (dict(resolution=256, class_cond=True, instance_cond=True, G_attn="64", D_attn="64",
                        G_ch=8, D_ch=8), 4, 1, False),
}
NS = 64
NS_GRAD = None      # samples per GRADIENT tensor (None = NS); make_golden_real_widths.py raises it to 4096, i.e. every gradient
                    # of <= 4096 elements is stored in full and the large ones at 4096 strided positions


def fingerprint(t, ns=None):
    ns = ns or NS
    t = t.detach().double().flatten()
    n = t.numel()
    stride = max(n // ns, 1)
    s = t[::stride][:ns]
    samp = np.zeros(ns)
    samp[: s.numel()] = s.numpy()
    return float(t.sum()), float((t * t).sum()), samp


def pack(d, ns=None):
    ns = ns or NS
    names = list(d.keys())
    fp = [fingerprint(d[k], ns) for k in names]
    return dict(names=json.dumps(names), sum=np.array([f[0] for f in fp]),
                sq=np.array([f[1] for f in fp]), samp=np.stack([f[2] for f in fp]) if fp else np.zeros((0, ns)))


def run_case(name, over, gb, steps, full, probe=True):
    cfg = dict(BASE)
    cfg.update(over)
    torch.manual_seed(0)
    G = RefBigGAN.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = RefBigGAN.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    gspec, dspec = synth.spec_of(G.state_dict()), synth.spec_of(D.state_dict())
    G.load_state_dict(synth.synth_state(gspec, seed=11))
    D.load_state_dict(synth.synth_state(dspec, seed=22))
    out = {"cfg": json.dumps(cfg), "gspec": json.dumps(gspec), "dspec": json.dumps(dspec),
           "dim_z": np.array(G.dim_z), "g_batch": np.array(gb), "steps": np.array(steps)}
    G.train(); D.train()

    # ---- forward-only probe on fresh weights (buffers are mutated: u0, stored_*) ----
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=5)
    c = samp()
    z = c[0] if isinstance(c, tuple) else c
    lab = c[1] if cfg["class_cond"] else None
    fg = (c[-1] if cfg["instance_cond"] else None)
    taps = {}
    hooks = []
    for i, bl in enumerate(G.blocks):
        hooks.append(bl[0].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"g.block{i}", o.detach())))
        if len(bl) > 1:
            hooks.append(bl[1].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"g.attn{i}", o.detach())))
    for i, bl in enumerate(D.blocks):
        hooks.append(bl[0].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"d.block{i}", o.detach())))
    with torch.no_grad():
        img = G(z, lab, fg)
        logit = D(img, lab, fg)
    for h in hooks:
        h.remove()
    if full:
        out["fwd/img"] = img.numpy()
    out["fwd/logit"] = logit.numpy()
    for k, v in pack({**taps, "img": img}).items():
        out["fwd/taps/" + k] = v
    for k, v in pack(G.state_dict()).items():
        out["fwd/G_state/" + k] = v
    for k, v in pack(D.state_dict()).items():
        out["fwd/D_state/" + k] = v

    # ---- training steps through the reference step function, from fresh weights ----
    G.load_state_dict(synth.synth_state(gspec, seed=11))
    D.load_state_dict(synth.synth_state(dspec, seed=22))
    G_ema = RefBigGAN.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = ref_utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]),
                             weight_decay=0, eps=cfg["adam_eps"])
    opt_g = torch.optim.Adam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]),
                             weight_decay=0, eps=cfg["adam_eps"])
    GD = RefBigGAN.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
    train = ref_train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp,
                                                embedded_optimizers=False, device="cpu", batch_size=gb)
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    losses = []
    for s in range(steps):
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        m = train(x, y, f)
        losses.append([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]])
        if s == 0:
            for k, v in pack({n: p.grad for n, p in G.named_parameters() if p.grad is not None}, NS_GRAD).items():
                out["step1/G_grad/" + k] = v
            for k, v in pack({n: p.grad for n, p in D.named_parameters() if p.grad is not None}, NS_GRAD).items():
                out["step1/D_grad/" + k] = v
        for k, v in pack(G.state_dict()).items():
            out[f"step{s + 1}/G_state/" + k] = v
        for k, v in pack(D.state_dict()).items():
            out[f"step{s + 1}/D_state/" + k] = v
        for k, v in pack(G_ema.state_dict()).items():
            out[f"step{s + 1}/EMA_state/" + k] = v
    out["losses"] = np.array(losses)
    path = os.path.join(HERE, f"biggan_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "losses", losses, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, (over, gb, steps, full) in CASES.items():
        if only and name not in only:
            continue
        run_case(name, over, gb, steps, full)

#!/usr/bin/env python
"""Goldens for the IC-GAN StyleGAN2 backbone from the UNMODIFIED reference on CPU:
training/networks.py (Generator, Discriminator), training/loss.py (StyleGAN2Loss.accumulate_gradients),
torch.optim.Adam, and the iteration structure of training/training_loop.py:313-346,428-531 (phases with lazy
regularisation, nan_to_num, Adam, G_ema) — the loop itself is one monolithic function in the reference, so the few lines
of glue around the reference's loss / networks / optimiser objects are restated here.

Recorded per configuration of tests/stylegan_cases.py::SG2_NETS:
  * state_dict specs of G and D                              (checkpoint-layout contract)
  * G(z,c,h) (train mode, noise_mode='const') and D logits   on synthetic weights
  * eval-mode G_ema-style sample with truncation_psi=0.7
  * parameter-gradient fingerprints of each phase Gmain / Greg / Dmain / Dreg
  * losses-free state fingerprints of G, D, G_ema after training iterations 1 (all phases) and 2 (main phases only)
Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stylegan2.py"""
import copy
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/stylegan2_ada_pytorch")
import numpy as np
import torch
from training import networks as ref_net
from training import loss as ref_loss

sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.stylegan_cases import SG2_LOSS, SG2_NETS, SG2_OPT, SG2_REAL_NETS, sg2_inputs, sg2_state   # noqa: E402

NS = 64


def fingerprint(t):
    t = t.detach().double().flatten()
    stride = max(t.numel() // NS, 1)
    s = t[::stride][:NS]
    samp = np.zeros(NS)
    samp[: s.numel()] = s.numpy()
    return float(t.sum()), float((t * t).sum()), samp


def pack(d):
    names = list(d.keys())
    fp = [fingerprint(d[k]) for k in names]
    return dict(names=json.dumps(names), sum=np.array([f[0] for f in fp]), sq=np.array([f[1] for f in fp]),
                samp=np.stack([f[2] for f in fp]) if fp else np.zeros((0, NS)))


def spec_of(m):
    return [(k, list(v.shape)) for k, v in m.state_dict().items()]


def load(m, seed):
    sd = sg2_state(spec_of(m), seed)
    cur = m.state_dict()
    m.load_state_dict({k: (cur[k] if v is None else v) for k, v in sd.items()})


def run(name, cfg):
    out = {}
    G = ref_net.Generator(**cfg["G"]).train().requires_grad_(False)
    D = ref_net.Discriminator(**cfg["D"]).train().requires_grad_(False)
    load(G, 1)
    load(D, 2)
    G_ema = copy.deepcopy(G).eval()
    out["gspec"], out["dspec"] = json.dumps(spec_of(G)), json.dumps(spec_of(D))
    b = cfg["batch"]
    z, gc, gh, img, rc, rh = sg2_inputs(cfg, 7, 4)

    with torch.no_grad():
        fake = G(z[:b], gc[:b], gh[:b], noise_mode="const")
        out["fwd/img"] = fake.numpy()          # (1.5 MB at 256 x 256, batch 2)
        out["fwd/logits_fake"] = D(fake, gc[:b], gh[:b]).numpy()
        out["fwd/logits_real"] = D(img, rc, rh).numpy()
        out["fwd/w_avg"] = G.mapping.w_avg.numpy().copy()
        G.eval()
        out["sample/img"] = G(z[:b], gc[:b], gh[:b], truncation_psi=0.7, noise_mode="const").numpy()
        G.train()
    load(G, 1)      # w_avg was updated by the training-mode forward

    # ---- per-phase gradients -------------------------------------------------------------------------------------
    for pi, phase in enumerate(["Gmain", "Greg", "Dmain", "Dreg"]):
        L = ref_loss.StyleGAN2Loss(device="cpu", G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, **SG2_LOSS)
        mod = G if phase[0] == "G" else D
        load(G, 1)
        mod.requires_grad_(True)
        for p in mod.parameters():
            p.grad = None
        torch.manual_seed(100 + pi)
        L.accumulate_gradients(phase=phase, real_img=img, real_c=rc, real_h=rh, gen_z=z[:b], gen_c=gc[:b], gen_h=gh[:b],
                               sync=True, gain=cfg.get("phase_gain", 1))
        mod.requires_grad_(False)
        grads = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in mod.named_parameters()}
        for k, v in pack(grads).items():
            out[f"grad/{phase}/{k}"] = v
        out[f"grad/{phase}/pl_mean"] = np.array(float(L.pl_mean))
        for p in mod.parameters():
            p.grad = None
    load(G, 1)

    # ---- two training iterations ---------------------------------------------------------------------------------
    L = ref_loss.StyleGAN2Loss(device="cpu", G_mapping=G.mapping, G_synthesis=G.synthesis, D=D, **SG2_LOSS)
    phases = []
    for pname, module, interval in (("G", G, 4), ("D", D, 16)):
        mb = interval / (interval + 1)
        opt = torch.optim.Adam(module.parameters(), lr=SG2_OPT["lr"] * mb, betas=[x ** mb for x in SG2_OPT["betas"]],
                               eps=SG2_OPT["eps"])
        phases += [(pname + "main", module, opt, 1), (pname + "reg", module, opt, interval)]
    cur_nimg, ema_kimg = 0, 0.02
    for it in range(2):
        z, gc, gh, img, rc, rh = sg2_inputs(cfg, 20 + it, len(phases))
        torch.manual_seed(500 + it)
        for i, (pname, module, opt, interval) in enumerate(phases):
            if it % interval != 0:
                continue
            opt.zero_grad(set_to_none=True)
            module.requires_grad_(True)
            sl = slice(i * b, (i + 1) * b)
            L.accumulate_gradients(phase=pname, real_img=img, real_c=rc, real_h=rh, gen_z=z[sl], gen_c=gc[sl],
                                   gen_h=gh[sl], sync=True, gain=interval)
            module.requires_grad_(False)
            for p in module.parameters():
                if p.grad is not None:
                    torch.nan_to_num(p.grad, nan=0, posinf=1e5, neginf=-1e5, out=p.grad)
            opt.step()
        with torch.no_grad():
            beta = 0.5 ** (b / max(ema_kimg * 1000, 1e-8))
            for p_ema, p in zip(G_ema.parameters(), G.parameters()):
                p_ema.copy_(p.lerp(p_ema, beta))
            for b_ema, bb in zip(G_ema.buffers(), G.buffers()):
                b_ema.copy_(bb)
        cur_nimg += b
        for tag, m in (("G", G), ("D", D), ("G_ema", G_ema)):
            for k, v in pack(m.state_dict()).items():
                out[f"iter{it + 1}/{tag}/{k}"] = v
        out[f"iter{it + 1}/pl_mean"] = np.array(float(L.pl_mean))
    path = os.path.join(HERE, f"stylegan2_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    import time
    # the real cfg4 network only on request (minutes of CPU time): python make_golden_stylegan2.py cfg4_r256 [cfg4_r256_fp16]
    for name, cfg in {**SG2_NETS, **SG2_REAL_NETS}.items():
        if (len(sys.argv) > 1 and name not in sys.argv[1:]) or (len(sys.argv) == 1 and name in SG2_REAL_NETS):
            continue
        t0 = time.time()
        run(name, cfg)
        print("%s: %.0f s" % (name, time.time() - t0), flush=True)

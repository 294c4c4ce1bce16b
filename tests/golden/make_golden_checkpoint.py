#!/usr/bin/env python
"""Writes a checkpoint with the REFERENCE's own `utils.save_weights` after two reference training steps, plus what
the reference produces from that checkpoint afterwards (tests/golden/ckpt_cc_ic_r64/ + ckpt_cc_ic_r64.npz):

  * files G.pth, D.pth, G_ema.pth, G_optim.pth, D_optim.pth, state_dict.pth            (utils.py:1116-1167)
  * eval-mode samples of G_ema and G for stored (z, y, feats) — the sampling path        (inference/utils.py:176-269)
  * losses + state fingerprints of training step 3 (resume: load -> one more step)

Runs only in the build container (needs /root/reference).  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py
"""
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up the reference import path + stubs)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import synth  # noqa: E402

NAME = "ckpt_cc_ic_r64"
CFG = dict(MG.BASE, resolution=64, class_cond=True, instance_cond=True, G_ch=4, D_ch=4, D_attn="16")   # small: the .pth files are committed
GB = 4


def main():
    cfg = dict(CFG)
    torch.manual_seed(0)
    G = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = MG.RefBigGAN.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    G.load_state_dict(synth.synth_state(synth.spec_of(G.state_dict()), seed=11))
    D.load_state_dict(synth.synth_state(synth.spec_of(D.state_dict()), seed=22))
    G_ema = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = MG.ref_utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    opt_g = torch.optim.Adam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    GD = MG.RefBigGAN.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0, "epoch": 0, "save_num": 0, "save_best_num": 0, "best_IS": 0, "best_FID": 999999,
             "config": {k: v for k, v in cfg.items()}}
    samp = synth.CondSampler(cfg, G.dim_z, GB, seed=7)
    train = MG.ref_train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False,
                                                   device="cpu", batch_size=GB)
    out = {"cfg": json.dumps(cfg), "dim_z": np.array(G.dim_z), "g_batch": np.array(GB)}

    def step(s):
        x, y, f = synth.synth_batch(cfg, GB, seed=100 + s)
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        m = train(x, y, f)
        return [m["G_loss"], m["D_loss_real"], m["D_loss_fake"]]

    out["losses_before"] = np.array([step(0), step(1)])

    root = os.path.join(HERE, NAME)
    shutil.rmtree(root, ignore_errors=True)
    MG.ref_utils.save_weights(G, D, state, HERE, NAME, None, G_ema, embedded_optimizers=False, G_optim=opt_g, D_optim=opt_d)

    # ---- sampling from the checkpointed weights, eval mode (stored BN statistics, no SN update) ----
    c = synth.CondSampler(cfg, G.dim_z, 6, seed=55)()
    z, lab, feats = c
    out["sample/z"], out["sample/y"], out["sample/feats"] = z.numpy(), lab.numpy(), feats.numpy()
    with torch.no_grad():
        G_ema.eval(); G.eval()
        out["sample/G_ema"] = G_ema(z, lab, feats).numpy()
        out["sample/G"] = G(z, lab, feats).numpy()
        D.eval()
        out["sample/D_logit"] = D(torch.from_numpy(out["sample/G_ema"]), lab, feats).numpy()
        G_ema.train(); G.train(); D.train()

    # ---- resume: one more step after the checkpoint ----
    out["losses_after"] = np.array([step(2)])
    for k, v in MG.pack(G.state_dict()).items():
        out["after/G_state/" + k] = v
    for k, v in MG.pack(D.state_dict()).items():
        out["after/D_state/" + k] = v
    for k, v in MG.pack(G_ema.state_dict()).items():
        out["after/EMA_state/" + k] = v
    path = os.path.join(HERE, NAME + ".npz")
    np.savez_compressed(path, **out)
    sizes = {f: os.path.getsize(os.path.join(root, f)) // 1024 for f in sorted(os.listdir(root))}
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", sizes)


if __name__ == "__main__":
    main()

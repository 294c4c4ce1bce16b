#!/usr/bin/env python
"""How far do the UNMODIFIED reference's own gradients move when its fp32 arithmetic is perturbed at the rounding level?
(build container only; needs /root/reference)

The reference train() call of a real-width golden case is run twice on the CPU: as is, and with every weight of G and D
multiplied by (1 + eps * N(0,1)), eps = 1e-6 -- the size of one layer's fp32 summation-order / Winograd rounding.  Printed per
gradient tensor (4096 strided samples, as in the goldens): rms and max of the difference in units of the tensor's rms, and
the fraction of samples that exceed GRAD_RTOL * rms.  A ReLU / max-pool network answers such a perturbation with a small rms
change and ISOLATED large changes (units whose pre-activation sits within rounding of zero flip their mask), i.e. with
max / rms ratios far above the ~4 of a Gaussian -- the same signature the HIP path shows against the goldens
(profiles/r03_parity_report.txt).  Usage: python tools/ref_sensitivity.py [case] [eps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np                       # noqa: E402
import torch                             # noqa: E402
import make_golden as MG                 # noqa: E402
import make_golden_real_widths as RW     # noqa: E402

synth = MG.synth
NS = 4096


def grads(cfg, gb, eps, seed):
    G = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = MG.RefBigGAN.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    gsd = synth.synth_state(synth.spec_of(G.state_dict()), 11)
    dsd = synth.synth_state(synth.spec_of(D.state_dict()), 22)
    if eps:
        g = torch.Generator().manual_seed(seed)
        for sd in (gsd, dsd):
            for k, v in sd.items():
                if k.endswith("weight") and v.is_floating_point():
                    v.mul_(1 + eps * torch.randn(v.shape, generator=g))
    G.load_state_dict(gsd); D.load_state_dict(dsd)
    G_ema = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = MG.ref_utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    od = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    og = torch.optim.Adam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    GD = MG.RefBigGAN.G_D(G, D, optimizer_G=og, optimizer_D=od)
    train = MG.ref_train_fns.GAN_training_function(G, D, GD, ema, {"itr": 1}, cfg, synth.CondSampler(cfg, G.dim_z, gb, seed=7),
                                                   embedded_optimizers=False, device="cpu", batch_size=gb)
    x, y, f = synth.synth_batch(cfg, gb, seed=100)
    G.train(); D.train(); G_ema.train()
    train(x, y, f)
    out = {}
    for tag, net in (("G ", G), ("D ", D)):
        for n, p in net.named_parameters():
            if p.grad is not None:
                out[tag + n] = MG.fingerprint(p.grad, NS)
    return out


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "cfg2_w96_r128"
    eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
    mk, gb, _, _ = RW.REAL_CASES[case]
    a, b = grads(mk(), gb, 0.0, 0), grads(mk(), gb, eps, 1)
    rows = []
    for k in a:
        n = max(int(np.count_nonzero(a[k][2])), 1)
        rms = np.sqrt(a[k][1] / max(len(a[k][2]), 1)) if False else np.sqrt((a[k][2] ** 2).sum() / n)
        d = b[k][2] - a[k][2]
        if rms < 1e-7:
            continue
        rows.append((k, n, rms, np.sqrt((d ** 2).sum() / n) / rms, np.abs(d).max() / rms, float((np.abs(d) > 1.5e-2 * rms).mean())))
    rows.sort(key=lambda r: -r[4])
    print("%s, batch %d: reference fp32 gradients, weights perturbed by %.0e (relative, Gaussian) vs unperturbed" % (case, gb, eps))
    for r in rows[:14]:
        print("  %-34s n=%5d rms %.2e   diff rms %.2e  max %.2e  (max/rms %5.1f)  > GRAD_RTOL*rms: %.4f" % (r + (0,))[:7] if False else
              "  %-34s n=%5d rms %.2e   diff rms %.2e  max %.2e  (max/rms %5.1f)  share > GRAD_RTOL*rms: %.4f" % (r[0], r[1], r[2], r[3], r[4], r[4] / max(r[3], 1e-30), r[5]))
    print("  median over %d tensors: diff rms %.2e, max %.2e, max/rms %.1f" % (len(rows), np.median([r[3] for r in rows]),
          np.median([r[4] for r in rows]), np.median([r[4] / max(r[3], 1e-30) for r in rows])))

#!/usr/bin/env python
"""Times the UNMODIFIED reference (BigGAN_PyTorch train_fns.GAN_training_function on CPU) on the build container's cores:
cfg1 at its full batch 8 and cfg3 at batch 2 (SURVEY 8(d) "CPU baseline": reduced batch, scales ~linearly), plus the oracle
port (oracle.biggan_oracle.train_step) on the same inputs, so that the port bench.py times on the GPU box's host can be compared
with the thing it restates.  Build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tools/ref_cpu_timing.py  ->  lines for BASELINE.md
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as MG                 # noqa: E402  (imports the reference with the torchvision stub)
import make_golden_real_widths as RW     # noqa: E402
import torch                             # noqa: E402

synth = MG.synth


def time_reference(cfg, gb, steps=3):
    G = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = MG.RefBigGAN.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    gspec, dspec = synth.spec_of(G.state_dict()), synth.spec_of(D.state_dict())
    G.load_state_dict(synth.synth_state(gspec, seed=11)); D.load_state_dict(synth.synth_state(dspec, seed=22))
    G_ema = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = MG.ref_utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    od = torch.optim.Adam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    og = torch.optim.Adam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), weight_decay=0, eps=cfg["adam_eps"])
    GD = MG.RefBigGAN.G_D(G, D, optimizer_G=og, optimizer_D=od)
    train = MG.ref_train_fns.GAN_training_function(G, D, GD, ema, {"itr": 1}, cfg, synth.CondSampler(cfg, G.dim_z, gb, seed=7),
                                                   embedded_optimizers=False, device="cpu", batch_size=gb)
    x, y, f = synth.synth_batch(cfg, gb, seed=100)
    G.train(); D.train(); G_ema.train()
    train(x, y, f)                       # warm-up
    ts = []
    for _ in range(steps):
        t0 = time.time(); train(x, y, f); ts.append(time.time() - t0)
    return sorted(ts)[len(ts) // 2], gspec, dspec


def time_port(cfg, gb, gspec, dspec, steps=3):
    from oracle import biggan_oracle as O
    gsd, dsd = synth.synth_state(gspec, 11), synth.synth_state(dspec, 22)
    ema_sd = {k: v.clone() for k, v in gsd.items()}
    og = O.AdamState(O.param_names(gsd), cfg["G_lr"], cfg["G_B1"], cfg["G_B2"], cfg["adam_eps"])
    od = O.AdamState(O.param_names(dsd), cfg["D_lr"], cfg["D_B1"], cfg["D_B2"], cfg["adam_eps"])
    x, y, f = synth.synth_batch(cfg, gb, seed=100)
    samp = synth.CondSampler(cfg, int(cfg.get("dim_z", 120)), gb, seed=7)
    ts = []
    for i in range(steps + 1):
        t0 = time.time(); O.train_step(gsd, dsd, ema_sd, cfg, og, od, x, y, f, samp, 1 + i, gb); ts.append(time.time() - t0)
    ts = ts[1:]
    return sorted(ts)[len(ts) // 2]


if __name__ == "__main__":
    n = torch.get_num_threads()
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    print("host: %d threads (%s), torch %s" % (n, cpu, torch.__version__))
    for name, gb in (("cfg1_icgan_res64", 8), ("cfg3_w96_r256", 2)):
        mk = RW.REAL_CASES[name][0]
        cfg = mk()
        t_ref, gspec, dspec = time_reference(cfg, gb)
        dz = MG.RefBigGAN.Generator(**{**cfg, "skip_init": True, "no_optim": True}).dim_z
        t_port = time_port({**cfg, "dim_z": dz}, gb, gspec, dspec)
        print("%-18s batch %d: unmodified reference %.2f s/step = %.3f images/s | oracle port %.2f s/step = %.3f images/s (port/reference time %.2f)"
              % (name, gb, t_ref, gb / t_ref, t_port, gb / t_port, t_port / t_ref), flush=True)

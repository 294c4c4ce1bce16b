#!/usr/bin/env python
"""Which ATen operations does the BigGAN training step issue OUTSIDE the C-ABI calls?  (tools only, CPU)

Runs one step of a small golden case with the kernels emulated (oracle/kernel_ref.py, as the CPU host-logic tests do) under a
TorchDispatchMode that is switched off while an emulated kernel runs, and prints every ATen op the product's own Python issued,
with the innermost ic_gan_amd/ frame that asked for it.  On the GPU each line is a kernel launch (or a rocclr copy / fill) of the
step's launch tail (`profiles/r04_bench_cfg3_kernel_stats.csv`: the at::native / __amd_rocclr rows)."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from oracle import kernel_ref, synth
from tests.helpers import load_golden

CHEAP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute", "aten.expand",
         "aten.as_strided", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten.empty",
         "aten.split", "aten.unbind", "aten._local_scalar_dense", "aten.is_", "aten.lift_fresh", "aten.chunk", "aten.narrow")


class Patch:
    def setattr(self, obj, name, val):
        setattr(obj, name, val)


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.off = 0
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not self.off and not name.startswith(CHEAP):
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/ic_gan_amd/" in fr.filename and "_python_dispatch" not in fr.filename:
                    site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                    break
            big = max([a.numel() for a in list(args) + [out] if isinstance(a, torch.Tensor)] or [0])
            self.rows[(name, site, "big" if big > 4096 else "small")] += 1
        return out


def main(case="cc_ic_r64"):
    import ic_gan_amd._lib as L
    import ic_gan_amd.BigGAN as M
    import ic_gan_amd.ops as ops
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    kernel_ref.install(Patch())
    tr = Trace()
    for mod, names in ((L, ("call", "query")), (ops, ("adam_multi", "ema_multi", "sn_prepare_many"))):
        for n in names:
            f = getattr(mod, n)

            def wrapped(*a, _f=f, **k):
                tr.off += 1
                try:
                    return _f(*a, **k)
                finally:
                    tr.off -= 1
            setattr(mod, n, wrapped)
    g = load_golden(case)
    cfg = g["cfg"]
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    D.load_state_dict(synth.synth_state(g["dspec"], 22))
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    gb = int(g["g_batch"])
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False, device="cpu", batch_size=gb)
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    for s in range(2):                     # step 0 warms the caches (SN layouts, Adam state); step 1 is counted
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        if s == 1:
            with tr:
                train(x, y, f)
        else:
            train(x, y, f)
    total = sum(tr.rows.values())
    print("%s: %d ATen operations outside the C-ABI calls in one step (views / empty / item left out)" % (case, total))
    for (name, site, size), n in sorted(tr.rows.items(), key=lambda kv: -kv[1]):
        print("%5d  %-34s %-5s %s" % (n, name, size, site))


def main_sg2(case="ic_r32_fp16"):
    """the same for one StyleGAN2 training iteration (Gmain + Dmain: the iteration without the lazy regularisers)"""
    import copy
    import ic_gan_amd._lib as L
    import ic_gan_amd.ops as ops
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan2.training_step import TrainingStep
    from tests.stylegan_cases import SG2_NETS, SG2_LOSS, SG2_OPT, sg2_inputs, sg2_state
    from tests import stylegan_cases as SC
    kernel_ref.install(Patch())
    tr = Trace()
    for mod, names in ((L, ("call", "query")), (ops, ("adam_multi", "ema_multi", "nan_to_num_multi", "sg2_weight_prep_multi"))):
        for n in names:
            f = getattr(mod, n)

            def wrapped(*a, _f=f, **k):
                tr.off += 1
                try:
                    return _f(*a, **k)
                finally:
                    tr.off -= 1
            setattr(mod, n, wrapped)
    nets = dict(SG2_NETS)
    nets.update(getattr(SC, "SG2_REAL_NETS", {}))
    cfg = nets[case]
    G = N.Generator(**cfg["G"]).train().requires_grad_(False)
    D = N.Discriminator(**cfg["D"]).train().requires_grad_(False)
    for m, seed in ((G, 1), (D, 2)):
        sd = sg2_state([[k, list(v.shape)] for k, v in m.state_dict().items()], seed)
        cur = m.state_dict()
        m.load_state_dict({k: (cur[k] if v is None else v) for k, v in sd.items()})
    G_ema = copy.deepcopy(G).eval()
    b = cfg["batch"]
    step = TrainingStep(G, D, G_ema, "cpu", batch_size=b, batch_gpu=b, loss_kwargs=SG2_LOSS, G_opt_kwargs=SG2_OPT, D_opt_kwargs=SG2_OPT,
                        G_reg_interval=4, D_reg_interval=16, ema_kimg=0.02)
    for it in range(2):                    # iteration 0 runs the regularisers too; iteration 1 (Gmain + Dmain) is counted
        z, gc, gh, img, rc, rh = sg2_inputs(cfg, 20 + it, 4)
        if it == 1:
            with tr:
                ran = step(img, rc, rh, z, gc, gh)
        else:
            ran = step(img, rc, rh, z, gc, gh)
    total = sum(tr.rows.values())
    print("%s %s: %d ATen operations outside the C-ABI calls in one iteration" % (case, ran, total))
    by_site = collections.Counter()
    for (name, site, size), n in tr.rows.items():
        by_site[site] += n
    print("-- by call site")
    for site, n in by_site.most_common(40):
        print("%5d  %s" % (n, site))
    print("-- operations of the five largest sites")
    for site, n in by_site.most_common(5):
        ops_here = collections.Counter()
        for (name, st, size), k in tr.rows.items():
            if st == site:
                ops_here[name + " " + size] += k
        print("  %s: %s" % (site, ", ".join("%s x%d" % kv for kv in ops_here.most_common(14))))
    print("-- by operation")
    by_op = collections.Counter()
    for (name, site, size), n in tr.rows.items():
        by_op[name] += n
    for name, n in by_op.most_common(30):
        print("%5d  %s" % (n, name))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sg2":
        main_sg2(*sys.argv[2:])
    else:
        main(*sys.argv[1:])

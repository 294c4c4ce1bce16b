"""Is the cfg3 train step reproducible from process to process?  (round 6: two identical `bench.py` runs printed different losses after
13 steps.)  Builds the bench's cfg3 setup, runs K steps and prints after every step the three losses (repr) and an md5 over all
parameters of G and D -- run it twice (or more) and diff the output.  `--sync` puts a torch.cuda.synchronize() after every C-ABI call
(host/device ordering taken out); `--workload cfg2` the 128x128 network.
    python tools/determinism_cfg3.py [steps] [--sync] [--batch B]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
    wl = sys.argv[sys.argv.index("--workload") + 1] if "--workload" in sys.argv else "cfg3"
    over, batch = bench.WORKLOADS[wl]
    if "--batch" in sys.argv:
        batch = int(sys.argv[sys.argv.index("--batch") + 1])
    cfg = dict(bench.BASE_CFG)
    cfg.update(over)
    cfg["num_D_accumulations"] = cfg["num_G_accumulations"] = 1
    from ic_gan_amd import train_fns, utils
    import ic_gan_amd._lib as L
    if "--sync" in sys.argv:
        orig = L.call
        L.call = lambda name, *a: (orig(name, *a), torch.cuda.synchronize())[0]
    device = "cuda:0"
    utils.seed_rng(0)
    M, G, D, G_ema, ema, opt_g, opt_d, init = bench.build_models(cfg, device, "N02")
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    sampler = bench.conditioning_sampler(cfg, G.dim_z, batch, device, seed=1000)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, sampler, embedded_optimizers=False, device=device, batch_size=batch)
    x, y, f = bench.synthetic_batch(cfg, batch, seed=7)
    x, y, f = x.to(device), (y.to(device) if y is not None else None), (f.to(device) if f is not None else None)

    def digest(net):
        h = hashlib.md5()
        for p in net.parameters():
            h.update(p.detach().cpu().numpy().tobytes())
        return h.hexdigest()[:12]

    print("init", digest(G), digest(D), flush=True)
    for s in range(steps):
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        m = train(x, y, f)
        torch.cuda.synchronize()
        print("step", s + 1, repr(m["G_loss"]), repr(m["D_loss_real"]), repr(m["D_loss_fake"]), "G", digest(G), "D", digest(D), flush=True)
        if "--grads" in sys.argv:
            for tag, net in (("D", D), ("G", G)):
                for n, p in net.named_parameters():
                    if p.grad is not None:
                        print("grad", s + 1, tag, n, hashlib.md5(p.grad.detach().cpu().numpy().tobytes()).hexdigest()[:10], flush=True)


if __name__ == "__main__":
    main()

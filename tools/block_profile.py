#!/usr/bin/env python
"""Where the cfg3 training step spends its GPU time, per network block (tools only; not the product).

HIP events on the launch stream at every GBlock / DBlock / Attention boundary of the forward pass and -- through identity autograd
nodes on a block's input and output -- of the backward pass.  Prints, per block and phase of the step (D step / G step), the forward
and backward milliseconds, and what is left outside the blocks (spectral norm, Adam, EMA, losses, stem / head layers).

    python tools/block_profile.py [--steps 3] [--workload cfg3]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench


class _Mark(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, sink, key):
        ctx.sink, ctx.key = sink, key
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        ctx.sink.append((ctx.key, ev))
        return g, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg3")
    args = ap.parse_args()
    device = "cuda:0"
    over, batch = bench.WORKLOADS[args.workload]
    cfg = dict(bench.BASE_CFG)
    cfg.update(over)
    from ic_gan_amd import layers, train_fns, utils
    utils.seed_rng(0)
    M, G, D, G_ema, ema, opt_g, opt_d, _ = bench.build_models(cfg, device, "N02")
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    sampler = bench.conditioning_sampler(cfg, G.dim_z, batch, device, seed=1000)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, sampler, embedded_optimizers=False, device=device,
                                            batch_size=batch)
    x, y, f = bench.synthetic_batch(cfg, batch, seed=7)
    x, y, f = x.to(device), (y.to(device) if y is not None else None), (f.to(device) if f is not None else None)

    names = {}
    for tag, net in (("G", G), ("D", D)):
        for n, m in net.named_modules():
            if isinstance(m, (layers.GBlock, layers.DBlock, layers.Attention)):
                names[id(m)] = "%s.%s %s" % (tag, n, type(m).__name__)
    events = []          # (key, event) in stream order
    live = {"on": False}

    def wrap(cls):
        inner = cls.forward

        def forward(self, x, *a, **k):
            if not live["on"]:
                return inner(self, x, *a, **k)
            name = names[id(self)]
            if x.requires_grad:
                x = _Mark.apply(x, events, (name, "bwd_end"))
            e0 = torch.cuda.Event(enable_timing=True); e0.record(); events.append(((name, "fwd_start"), e0))
            out = inner(self, x, *a, **k)
            e1 = torch.cuda.Event(enable_timing=True); e1.record(); events.append(((name, "fwd_end"), e1))
            if out.requires_grad:
                out = _Mark.apply(out, events, (name, "bwd_start"))
            return out
        cls.forward = forward

    for cls in (layers.GBlock, layers.DBlock, layers.Attention):
        wrap(cls)

    def step():
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        return train(x, y, f)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    acc = collections.OrderedDict()
    total = 0.0
    for _ in range(args.steps):
        events.clear()
        live["on"] = True
        s0 = torch.cuda.Event(enable_timing=True); s0.record()
        step()
        s1 = torch.cuda.Event(enable_timing=True); s1.record()
        live["on"] = False
        torch.cuda.synchronize()
        total += s0.elapsed_time(s1)
        # a block runs several times per step (D step: G forward without grad, D forward + backward; G step: G and D forward + backward):
        # pair start / end events in stream order and number the occurrences
        open_, seen = {}, collections.Counter()
        last_bwd_start = {}
        for (name, what), ev in events:
            if what in ("fwd_start", "bwd_start"):
                open_[(name, what[:3])] = ev
                if what == "bwd_start":
                    last_bwd_start[name] = ev
            else:
                st = open_.pop((name, what[:3]), None)
                if st is None:
                    continue
                seen[(name, what[:3])] += 1
                key = (name, what[:3], seen[(name, what[:3])])
                acc[key] = acc.get(key, 0.0) + st.elapsed_time(ev)
        for (name, kind), st in open_.items():       # backward of a block whose input needs no gradient: runs to the end of that backward
            if kind == "bwd":
                key = (name, "bwd(open)", 1)
                acc[key] = acc.get(key, 0.0)
    n = args.steps
    inside = 0.0
    per_block = collections.OrderedDict()
    for (name, kind, occ), ms in acc.items():
        per_block.setdefault(name, []).append("%s#%d %.2f" % (kind, occ, ms / n))
        inside += ms / n
    print("step %.2f ms;  inside blocks %.2f ms;  outside (SN / Adam / EMA / stem / head / losses) %.2f ms" % (total / n, inside, total / n - inside))
    for name, items in per_block.items():
        tot = sum(float(i.split()[-1]) for i in items)
        print("%-34s %7.2f ms   %s" % (name, tot, "  ".join(items)))


if __name__ == "__main__":
    main()

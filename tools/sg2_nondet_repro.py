"""Round-6 item 1(a): loop the second-order SynthesisLayer check of tests/test_sg2_fused_gpu.py and record, per repetition, a hash of
every result of the composed path (g0) and of the fused path (g1), so that a moving result names the path that moved.
    python tools/sg2_nondet_repro.py [reps] [--poison]"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def h(t):
    return hashlib.md5(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:10]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 100
    from tests import test_sg2_fused_gpu as T
    if "--poison" in sys.argv:
        from tests import _poison
        _poison.install(sys.argv[sys.argv.index("--poison") + 1])
    from ic_gan_amd.stylegan2 import networks as N
    cases = [(512, 512, 16, 2, False, "const", 4), (512, 512, 16, 1, False, "random", 4), (512, 512, 32, 2, True, "random", 4)]
    for cin, cout, res, up, half, noise_mode, n in cases:
        layer = N.SynthesisLayer(cin, cout, w_dim=512, resolution=res, up=up, conv_clamp=256).cuda()
        T._init(layer, 3)
        draws = T.rnd(n, 1, res, res, seed=77).cuda()
        N._randn = lambda shape, device: draws.clone()
        x = T.rnd(n, cin, res // up, res // up, seed=5).cuda()
        w = T.rnd(n, 3, 512, seed=6).cuda()[:, 1]
        if half:
            x = x.half()
        fn = lambda x, w: layer(x, w, noise_mode=noise_mode, fused_modconv=False)
        params = list(layer.parameters())
        seen = {0: {}, 1: {}}
        first = {}
        for rep in range(reps):
            for fused in (False, True):
                g, gi, gp = T._second_order(fn, params, [x, w], fused)
                torch.cuda.synchronize()
                sig = (h(g),) + tuple(h(t) for t in gi) + tuple(h(t) if t is not None else "-" for t in gp)
                if sig not in seen[fused]:
                    seen[fused][sig] = rep
                    if fused not in first:
                        first[fused] = (g.clone(), [t.clone() for t in gi])
                    else:
                        g_f, gi_f = first[fused]
                        d = (g - g_f).abs()
                        print("case", (cin, cout, res, up, half), "fused" if fused else "composed", "rep", rep, "NEW signature", sig,
                              "| g max diff %.3e per-sample" % float(d.max()), [float(v) for v in d.flatten(1).max(1).values],
                              "| inputs diff", [float((a - b).abs().max()) for a, b in zip(gi, gi_f)], flush=True)
            if rep % 50 == 49:
                print("case", (cin, cout, res, up, half), "rep", rep + 1, "distinct composed", len(seen[0]), "distinct fused", len(seen[1]), flush=True)
        g0 = T._second_order(fn, params, [x, w], False)[0]
        g1 = T._second_order(fn, params, [x, w], True)[0]
        print("case", (cin, cout, res, up, half), "signatures composed", list(seen[0]), "fused", list(seen[1]))
        print("case", (cin, cout, res, up, half), "DONE distinct composed", len(seen[0]), "fused", len(seen[1]),
              "fused-vs-composed rel %.3e" % float((g1 - g0).abs().max() / g0.abs().max()), flush=True)




def sweep(k):
    """`--sweep K`: the layer's `noise_const` buffer (torch.randn at construction: the one tensor tests/test_sg2_fused_gpu.py::_init did not
    seed before round 6; torch's default seed differs per process in this build) drawn from K seeds -> fused-vs-composed error of the
    path-length vector per seed, and for each seed how many activations sit on opposite sides of the lrelu kink / the clamp in the two paths."""
    from tests import test_sg2_fused_gpu as T
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    cin, cout, res, up, n = 512, 512, 16, 2, 4
    layer = N.SynthesisLayer(cin, cout, w_dim=512, resolution=res, up=up, conv_clamp=256).cuda()
    T._init(layer, 3)
    x = T.rnd(n, cin, res // up, res // up, seed=5).cuda()
    w = T.rnd(n, 3, 512, seed=6).cuda()[:, 1]
    fn = lambda x, w: layer(x, w, noise_mode="const", fused_modconv=False)
    params = list(layer.parameters())
    rows = []
    seeds = [int(v) for v in sys.argv[sys.argv.index("--seeds") + 1].split(",")] if "--seeds" in sys.argv else None
    for seed in (seeds if seeds else range(k)):
        with torch.no_grad():
            layer.noise_const.copy_(T.rnd(res, res, seed=1000 + seed))
        # (with autograd on: without it both calls take the fused kernels)
        y0 = fn(x.clone().requires_grad_(True), w.clone().requires_grad_(True)).detach()
        with FL.second_order():
            y1 = fn(x.clone().requires_grad_(True), w.clone().requires_grad_(True)).detach()
        flips = int(((y0 > 0) != (y1 > 0)).sum()) + int(((y0.abs() < 256) != (y1.abs() < 256)).sum())
        g0 = T._second_order(fn, params, [x, w], False)[0]
        g1 = T._second_order(fn, params, [x, w], True)[0]
        per = ((g1 - g0).abs().flatten(1).max(1).values / g0.abs().max()).tolist()
        rows.append((max(per), seed, flips, per))
        print("noise seed %3d  rel err %.3e  per sample %s  kink flips %d  forward rel %.2e" % (
            seed, max(per), ["%.1e" % v for v in per], flips, float((y1 - y0).abs().max() / y0.abs().max())), flush=True)
        if flips:
            idx = ((y0 > 0) != (y1 > 0)).nonzero()
            for i in idx[:4].tolist():
                print("    flipped activation at [n, o, h, w] = %s: composed y = %.3e, fused y = %.3e" % (i, float(y0[tuple(i)]), float(y1[tuple(i)])))
    errs = sorted(r[0] for r in rows)
    k = len(errs)
    print("sweep of %d: median %.3e  p90 %.3e  max %.3e; seeds over 5e-5: %s; of those with kink flips: %s" % (
        k, errs[k // 2], errs[int(k * 0.9)], errs[-1], [r[1] for r in rows if r[0] > 5e-5], [r[1] for r in rows if r[0] > 5e-5 and r[2] > 0]))


if __name__ == "__main__":
    if "--sweep" in sys.argv:
        sweep(int(sys.argv[sys.argv.index("--sweep") + 1]))
    else:
        main()

"""Diagnostic (not collected by pytest): how far the ch=32 network's gradients move between the direct, F(2x2,3x3) and
F(4x4,3x3) forms, measured against an fp64 run of the CPU oracle (and the fp32 oracle beside it for scale).
    python tests/diag_winograd_accuracy.py          (needs the GPU)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, biggan_oracle as O
from tests.test_parity_gpu import WIDE, _build, _cond, _d, rel_l2
import ic_gan_amd.ops as ops


def oracle(cfg, gsd, dsd, z, lab, fg, x, y, f, B, dt):
    c = lambda t: t.to(dt) if t.is_floating_point() else t
    gsd = {k: c(v).clone() for k, v in gsd.items()}
    dsd = {k: c(v).clone() for k, v in dsd.items()}
    for k in O.param_names(gsd):
        gsd[k].requires_grad_(True)
    for k in O.param_names(dsd):
        dsd[k].requires_grad_(True)
    img = O.generator_forward(gsd, cfg, c(z), lab, c(fg), True)
    out = O.discriminator_forward(dsd, cfg, torch.cat([img, c(x)], 0), torch.cat([lab, y]), torch.cat([c(fg), c(f)]), True)
    (out[:B].mean() - 0.5 * out[B:].mean()).backward()
    return img.detach(), out.detach(), {k: v.grad for k, v in gsd.items() if v.grad is not None}, {k: v.grad for k, v in dsd.items() if v.grad is not None}


def main():
    cfg = dict(WIDE)
    B = 6
    _, G0, D0, gspec, dspec = _build(cfg)
    gsd, dsd = synth.synth_state(gspec, 11), synth.synth_state(dspec, 22)
    z, lab, fg = _cond(cfg, G0.dim_z, B, 3)
    x, y, f = synth.synth_batch(cfg, B, seed=9)
    i64, o64, gg64, dg64 = oracle(cfg, gsd, dsd, z, lab, fg, x, y, f, B, torch.float64)
    i32, o32, gg32, dg32 = oracle(cfg, gsd, dsd, z, lab, fg, x, y, f, B, torch.float32)
    top = max(float(v.norm()) for v in gg64.values())

    def report(tag, img, out, gg, dg):
        worst = sorted(((rel_l2(v.double().cpu(), gg64[n]), "G." + n) for n, v in gg.items() if float(gg64[n].norm()) >= 1e-4 * top), reverse=True)
        worst += sorted(((rel_l2(v.double().cpu(), dg64[n]), "D." + n) for n, v in dg.items()), reverse=True)[:3]
        worst.sort(reverse=True)
        print(f"{tag:12s} img {rel_l2(img.double().cpu(), i64):.2e} out {rel_l2(out.double().cpu(), o64):.2e}  worst grads: "
              + ", ".join(f"{n} {e:.2e}" for e, n in worst[:4]))

    report("oracle32", i32, o32, gg32, dg32)
    BIG = 10 ** 9
    off, allrs = {True: (BIG,) * 3, False: (BIG,) * 3}, {True: (4, 4, 4), False: (4, 4, 4)}
    prod = (ops.WINOGRAD_MIN_CHANNELS, ops.WINOGRAD2_MIN_CHANNELS, ops.WINOGRAD4_MIN_CHANNELS, ops.WINOGRAD4_WGRAD_MIN_CHANNELS,
            ops.RS_WINOGRAD_MIN_CHANNELS)
    for tag, knobs in (("no winograd", (BIG, BIG, BIG, BIG, off)), ("F(2,3)>=192", (192, 192, BIG, BIG, off)),
                       ("F(4,3) prod", prod[:4] + (off,)), ("product", prod),
                       ("F(2,3)all", (4, 4, BIG, BIG, off)), ("F(4,3)all", (4, 4, 4, 4, off)), ("all+rs all", (4, 4, 4, 4, allrs))):
        (ops.WINOGRAD_MIN_CHANNELS, ops.WINOGRAD2_MIN_CHANNELS, ops.WINOGRAD4_MIN_CHANNELS, ops.WINOGRAD4_WGRAD_MIN_CHANNELS,
         ops.RS_WINOGRAD_MIN_CHANNELS) = knobs
        _, G, D, _, _ = _build(cfg)
        G.train(); D.train()
        img = G(_d(z), _d(lab), _d(fg))
        d_in = torch.cat([img, _d(x).contiguous(memory_format=torch.channels_last)], 0)
        out = D(d_in, torch.cat([_d(lab), _d(y)]), torch.cat([_d(fg), _d(f)]))
        (out[:B].mean() - 0.5 * out[B:].mean()).backward()
        report(tag, img.detach(), out.detach(), {n: p.grad for n, p in G.named_parameters() if p.grad is not None},
               {n: p.grad for n, p in D.named_parameters() if p.grad is not None})


if __name__ == "__main__":
    main()

"""GPU parity of the whole hot path: the ic_gan_amd modules (HIP kernels through the C-ABI) against
  (1) the committed reference-generated goldens (tests/golden/*.npz),
  (2) the CPU oracle run live on the same seeded inputs at a wider configuration,
  (3) size-independent properties at BASELINE.json's full sizes (adjoint identities of the convolution
      triplet, normalisation statistics, value range / finiteness of a full cfg3-shaped step).
Tolerances: forward activations / losses ~1e-4 relative (north_star: samples within 1e-3 rel L2);
gradients and post-step parameters are compared through fingerprints at tests/helpers.py's GRAD_RTOL / STATE_RTOL of the
tensor rms (+ the conditioning slack of the real-width cases; the dense 4096-sample groups by check_group's four-part rule)."""
import json

import numpy as np
import pytest
import torch

from oracle import biggan_oracle as O
from oracle import synth
from tests.helpers import (BENCH_CASES, CASES, GRAD_RTOL, REAL_CASES, STATE_RTOL, adam_slack, check_group, conditioning_slack,
                           fingerprint, load_golden)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(cfg, gspec=None, dspec=None):
    import ic_gan_amd.BigGAN as M
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    gspec = gspec or synth.spec_of(G.state_dict())
    dspec = dspec or synth.spec_of(D.state_dict())
    G.load_state_dict(synth.synth_state(gspec, 11))
    D.load_state_dict(synth.synth_state(dspec, 22))
    return M, G.to(DEV), D.to(DEV), gspec, dspec


def _cond(cfg, dim_z, gb, seed):
    c = synth.CondSampler(cfg, dim_z, gb, seed=seed)()
    z = c[0] if isinstance(c, tuple) else c
    lab = c[1] if cfg["class_cond"] else None
    fg = c[-1] if cfg["instance_cond"] else None
    return z, lab, fg


def _d(t):
    return None if t is None else t.to(DEV)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("case", CASES + REAL_CASES + BENCH_CASES)
def test_forward_vs_golden(case):
    g = load_golden(case)
    cfg = g["cfg"]
    _, G, D, _, _ = _build(cfg, g["gspec"], g["dspec"])
    G.train(); D.train()
    z, lab, fg = _cond(cfg, G.dim_z, int(g["g_batch"]), 5)
    with torch.no_grad():
        img = G(_d(z), _d(lab), _d(fg))
        logit = D(img, _d(lab), _d(fg))
    assert img.shape == (int(g["g_batch"]), 3, cfg["resolution"], cfg["resolution"])
    if "fwd/img" in g:
        assert rel_l2(img, torch.from_numpy(g["fwd/img"])) < 1e-4          # north_star bound is 1e-3
        np.testing.assert_allclose(img.cpu().numpy(), g["fwd/img"], rtol=5e-4, atol=1e-4)
    # every case (also the fingerprint-only ones at real widths): the image's 64 strided samples and its rms against the
    # reference's -- rel. L2 over the samples < 1e-3 is north_star's sample-parity bound, we hold 2e-4
    names = json.loads(str(g["fwd/taps/names"]))
    gi = names.index("img")
    _, sq, samp = fingerprint(img)
    gsamp = g["fwd/taps/samp"][gi]
    rel = float(np.linalg.norm(samp - gsamp) / np.linalg.norm(gsamp))
    assert rel < 2e-4, ("image samples rel-L2", rel)
    assert abs(np.sqrt(sq) - np.sqrt(g["fwd/taps/sq"][gi])) <= 2e-4 * np.sqrt(g["fwd/taps/sq"][gi])
    scale = max(1.0, float(np.abs(g["fwd/logit"]).max()))
    np.testing.assert_allclose(logit.cpu().numpy(), g["fwd/logit"], rtol=5e-4, atol=5e-4 * scale)
    check_group(g, "fwd/G_state/", {k: v.cpu() for k, v in G.state_dict().items()}, 2e-4, 1e-6, "G buf ")
    check_group(g, "fwd/D_state/", {k: v.cpu() for k, v in D.state_dict().items()}, 2e-4, 1e-6, "D buf ")


# Winograd variants of the golden train-step test.  The goldens come from the reference run on the CPU (direct convolution);
#  -1: every Winograd route off (implicit GEMM, 2x2-phase and 4x4-stride-2 forms only)
#   0: production thresholds (ops.WINOGRAD*_MIN_CHANNELS, ops.RS_WINOGRAD_MIN_CHANNELS: the 96..128-channel layers of these
#      test networks run in the F(4x4,3x3) / 25-plane domains)
#   2 / 4: F(2x2,3x3) / F(4x4,3x3) forced onto every eligible 3x3 stride-1 layer;  5: 4 + every resample-fused layer in the
#      25-plane domain in all three directions
# Round 1 needed multipliers of 3-5x on every Winograd variant (incl. production):
# the rounding was the single fp32 accumulation chain of the plane GEMMs (partial sums ~40x the result they cancel to in the
# output transform); with two-level accumulation (csrc/gemm_conv.hip, BLK) the worst error / tolerance ratio over all
# checked groups of these five goldens is 0.81 at production thresholds and with F(4x4,3x3) forced everywhere
# (profiles/r02_parity_report.txt; single-level: up to 12.97).
WINO_VARIANTS = [-1, 0, 2, 4, 5]
# -1 and 0 (what the product runs) are held to the plain tolerances.  The FORCED variants put the Winograd forms onto the
# 8 ... 128-channel layers of these goldens, far below the production thresholds (96 channels); their gradient tolerance is 2x
# (round 1: 5x): on these 8-channel networks single gradient tensors move between 0.3x and 1.4x of the plain tolerance when
# only the summation order of an unrelated kernel changes (profiles/r02_parity_report.txt).
FORCED_GRAD_MULT = {-1: 1.0, 0: 1.0, 2: 2.0, 4: 2.0, 5: 2.0}


def _set_winograd(monkeypatch, wino):
    import ic_gan_amd.ops as _ops
    BIG = 10 ** 9
    if wino == -1:
        for k in ("WINOGRAD_MIN_CHANNELS", "WINOGRAD2_MIN_CHANNELS", "WINOGRAD4_MIN_CHANNELS", "WINOGRAD4_WGRAD_MIN_CHANNELS"):
            monkeypatch.setattr(_ops, k, BIG)
        monkeypatch.setattr(_ops, "RS_WINOGRAD_MIN_CHANNELS", {True: (BIG, BIG, BIG), False: (BIG, BIG, BIG)})
    elif wino:
        monkeypatch.setattr(_ops, "WINOGRAD_MIN_CHANNELS", 4)
        monkeypatch.setattr(_ops, "WINOGRAD2_MIN_CHANNELS", 4)
        monkeypatch.setattr(_ops, "WINOGRAD4_MIN_CHANNELS", 4 if wino >= 4 else BIG)      # F(4x4,3x3) / F(2x2,3x3)
        monkeypatch.setattr(_ops, "WINOGRAD4_WGRAD_MIN_CHANNELS", 4 if wino >= 4 else BIG)
        if wino == 5:
            monkeypatch.setattr(_ops, "RS_WINOGRAD_MIN_CHANNELS", {True: (4, 4, 4), False: (4, 4, 4)})


@pytest.mark.parametrize("wino", WINO_VARIANTS)
@pytest.mark.parametrize("case", CASES)
def test_train_steps_vs_golden(case, wino, monkeypatch):
    _train_steps_case(case, wino, monkeypatch)


@pytest.mark.parametrize("wino", [-1, 0])
@pytest.mark.parametrize("case", REAL_CASES)
def test_train_steps_vs_golden_real_widths(case, wino, monkeypatch):
    """BASELINE.json's configurations at their real widths (cfg1 exactly; cfg2 / cfg3 at ch 96, batch 2) against the
    reference-generated goldens: the production kernel routes (wino = 0: F(4x4,3x3) from 96 channels, 25-plane
    resample-fused layers) and the Winograd-free route (-1).  Gradient tolerance per tensor = GRAD_RTOL * rms + COND_K x the
    fp32 reference's own distance from its fp64 run (helpers.conditioning_slack): at cfg3 the Winograd-FREE route sits at
    1.6 x (GRAD_RTOL * rms) on blocks.0.0.conv1.weight where the reference itself is 0.82 x away from fp64."""
    _train_steps_case(case, wino, monkeypatch)


@pytest.mark.parametrize("wino", [-1, 0])
@pytest.mark.parametrize("case", BENCH_CASES)
def test_train_steps_vs_golden_bench_config(case, wino, monkeypatch):
    """The configuration the metric is quoted on (cfg3: 256x256, ch 96, class + instance conditioning, attention at 64) at
    B = 16 and at the benchmark's own B = 64, one full train() call against the UNMODIFIED reference run on the CPU
    (tests/golden/make_golden_real_widths.py): losses, every gradient (4096 samples per tensor, tensors of <= 4096 elements in
    full), every post-step tensor of G, D and G_ema -- at the batch sizes where BN statistics, split-K slice counts, persistent
    tile runs and the attention maps take the routes the benchmark takes.  Production route (0) and Winograd-free route (-1).
    B = 16 has an fp64 twin of the reference (conditioning slack as for the batch-2 cases); B = 64 has none (the fp64 graph
    does not fit the build container) and is held to the plain GRAD_RTOL."""
    _train_steps_case(case, wino, monkeypatch)


def _train_steps_case(case, wino, monkeypatch, strict=True):
    _set_winograd(monkeypatch, wino)
    grad_rtol, state_rtol, slack_mult = GRAD_RTOL * FORCED_GRAD_MULT[wino], STATE_RTOL, 1.0
    # real-width cases carry an fp64 run of the reference: its distance from the fp32 goldens floors the gradient tolerance
    cond_g, cond_d = conditioning_slack(case, "step1/G_grad/"), conditioning_slack(case, "step1/D_grad/")
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    g = load_golden(case)
    cfg = g["cfg"]
    M, G, D, _, _ = _build(cfg, g["gspec"], g["dspec"])
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True}).to(DEV)
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    gb, steps = int(g["g_batch"]), int(g["steps"])
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False,
                                            device=DEV, batch_size=gb)
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}
    for s in range(steps):
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        m = train(_d(x), _d(y), _d(f))
        np.testing.assert_allclose([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]], g["losses"][s],
                                   rtol=1e-3, atol=1e-3)
        if s == 0:
            check_group(g, "step1/G_grad/", {n: p.grad.cpu() for n, p in G.named_parameters() if p.grad is not None},
                        grad_rtol, 1e-6, "G grad ", extra_atol=cond_g)
            check_group(g, "step1/D_grad/", {n: p.grad.cpu() for n, p in D.named_parameters() if p.grad is not None},
                        grad_rtol, 1e-6, "D grad ", extra_atol=cond_d)
        gx = {k: v * slack_mult for k, v in adam_slack(g, "step1/G_grad/", cfg["G_lr"], s + 1, G.state_dict().keys()).items()}
        dx = {k: v * slack_mult for k, v in adam_slack(g, "step1/D_grad/", cfg["D_lr"], s + 1, D.state_dict().keys()).items()}
        check_group(g, f"step{s + 1}/G_state/", cpu(G.state_dict()), state_rtol, 2e-6, "G ", extra_atol=gx)
        check_group(g, f"step{s + 1}/D_state/", cpu(D.state_dict()), state_rtol, 2e-6, "D ", extra_atol=dx)
        check_group(g, f"step{s + 1}/EMA_state/", cpu(G_ema.state_dict()), state_rtol, 2e-6, "EMA ", extra_atol=gx)


WIDE = dict(dim_z=120, shared_dim=128, shared_dim_feat=512, G_shared=True, G_shared_feat=True, hier=True,
            n_classes=100, SN_eps=1e-6, BN_eps=1e-5, G_ch=32, D_ch=32, G_attn="32", D_attn="32", resolution=64,
            class_cond=True, instance_cond=True)


def test_forward_backward_vs_live_oracle_wide():
    """ch=32 (channel counts 64..512, every vectorised kernel path, F(2x2,3x3) and F(4x4,3x3) Winograd at their production
    thresholds, the resample-fused layers in the 25-plane domain) against the CPU oracle run here in fp64: the fp32 oracle's
    own gradients sit 1.0e-3 (rel. L2) from that truth on this network, the HIP path without Winograd 1.4e-3, with the
    production routes 2.9e-3 -- tests/diag_winograd_accuracy.py prints the table; bound 5e-3."""
    cfg = dict(WIDE)
    _, G, D, gspec, dspec = _build(cfg)
    gsd, dsd = synth.synth_state(gspec, 11), synth.synth_state(dspec, 22)
    B = 6
    z, lab, fg = _cond(cfg, G.dim_z, B, 3)
    x, y, f = synth.synth_batch(cfg, B, seed=9)
    G.train(); D.train()
    # --- product
    img = G(_d(z), _d(lab), _d(fg))
    d_in = torch.cat([img, _d(x).contiguous(memory_format=torch.channels_last)], 0)
    out = D(d_in, torch.cat([_d(lab), _d(y)]), torch.cat([_d(fg), _d(f)]))
    loss = out[:B].mean() - 0.5 * out[B:].mean()
    loss.backward()
    # --- oracle (fp64)
    c64 = lambda t: t.double() if t.is_floating_point() else t
    gsd, dsd = {k: c64(v) for k, v in gsd.items()}, {k: c64(v) for k, v in dsd.items()}
    z, fg, x, f = z.double(), fg.double(), x.double(), f.double()
    for k in O.param_names(gsd):
        gsd[k].requires_grad_(True)
    for k in O.param_names(dsd):
        dsd[k].requires_grad_(True)
    img_o = O.generator_forward(gsd, cfg, z, lab, fg, True)
    out_o = O.discriminator_forward(dsd, cfg, torch.cat([img_o, x], 0), torch.cat([lab, y]), torch.cat([fg, f]), True)
    loss_o = out_o[:B].mean() - 0.5 * out_o[B:].mean()
    loss_o.backward()
    assert rel_l2(img, img_o) < 2e-4, rel_l2(img, img_o)
    assert rel_l2(out, out_o) < 5e-4, rel_l2(out, out_o)
    top = max(float(v.grad.norm()) for k, v in gsd.items() if v.grad is not None)
    for n, p in G.named_parameters():
        ref = gsd[n].grad
        if float(ref.norm()) < 1e-4 * top:
            continue                                   # mathematically-zero gradients (bias feeding BN)
        assert rel_l2(p.grad, ref) < 5e-3, (n, rel_l2(p.grad, ref))
    for n, p in D.named_parameters():
        assert rel_l2(p.grad, dsd[n].grad) < 5e-3, (n, rel_l2(p.grad, dsd[n].grad))
    for k, v in G.state_dict().items():
        if k.endswith(("u0", "sv0", "stored_mean", "stored_var")):
            assert rel_l2(v, gsd[k]) < 1e-4, k


def test_eval_mode_sampling_matches_oracle():
    """G.eval() (stored statistics, no power-iteration update) — the sampling path of G_ema."""
    cfg = dict(WIDE, G_ch=16, D_ch=16)
    _, G, D, gspec, _ = _build(cfg)
    gsd = synth.synth_state(gspec, 11)
    z, lab, fg = _cond(cfg, G.dim_z, 4, 21)
    G.eval()
    before = {k: v.clone() for k, v in G.state_dict().items()}
    with torch.no_grad():
        img = G(_d(z), _d(lab), _d(fg))
        img_o = O.generator_forward(gsd, cfg, z, lab, fg, False)
    assert rel_l2(img, img_o) < 2e-4
    for k, v in G.state_dict().items():
        assert torch.equal(v, before[k]), f"eval-mode forward mutated {k}"


# ------------------------------------------------------------------------------------------------ full-size properties
def test_conv_adjoint_identities_full_size():
    """<conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)> on the largest cfg3 layer shape
    (192 -> 96 @ 256x256 with fused affine+ReLU+upsample), batch 8: size-independent property."""
    import ic_gan_amd._lib as L
    B, Cin, Cout, Hs = 8, 192, 96, 128
    H = 2 * Hs
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(B, Cin, Hs, Hs, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, 3, 3, Cin, generator=g) / np.sqrt(9 * Cin)).to(DEV)
    dy = torch.randn(B, Cout, H, H, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    sc = (1 + 0.2 * torch.randn(B, Cin, generator=g)).to(DEV)
    sh = (0.2 * torch.randn(B, Cin, generator=g)).to(DEV)
    flags = L.ICG_PRE_AFFINE | L.ICG_PRE_RELU | L.ICG_UPSAMPLE2X
    out = torch.empty(B, Cout, H, H, device=DEV).contiguous(memory_format=torch.channels_last)
    L.call("icg_conv2d_fprop", x, w, None, None, out, sc, sh, Cin, B, H, H, Cin, Cout, 3, flags, 1.0)
    # a = act(x) upsampled, materialised by a 1x1 identity "conv" is avoided: use the linear-in-w identity
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, H, Cin, Cout, 3)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    dw = torch.empty(3, 3, Cin, Cout, device=DEV)
    L.call("icg_conv2d_wgrad", x, dy, dw, sc, sh, Cin, B, H, H, Cin, Cout, 3, flags, ws, nb)
    lhs = float((out.double() * dy.double()).sum())
    rhs_w = float((dw.double() * w.permute(1, 2, 3, 0).double()).sum())
    assert abs(lhs - rhs_w) <= 1e-6 * float(out.double().norm() * dy.double().norm()), (lhs, rhs_w)
    # dgrad adjoint on the un-fused conv (da is the gradient w.r.t. the conv input)
    x2 = torch.randn(B, Cin, H, H, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    out2 = torch.empty_like(out)
    L.call("icg_conv2d_fprop", x2, w, None, None, out2, None, None, 0, B, H, H, Cin, Cout, 3, 0, 1.0)
    wd = w.view(Cout, 3, 3, Cin).flip(1, 2).permute(3, 1, 2, 0).contiguous()
    da = torch.empty_like(x2)
    L.call("icg_conv2d_fprop", dy, wd, None, None, da, None, None, 0, B, H, H, Cout, Cin, 3, 0, 1.0)
    lhs2 = float((out2.double() * dy.double()).sum())
    rhs2 = float((x2.double() * da.double()).sum())
    scale = float(out2.double().norm() * dy.double().norm())
    assert abs(lhs2 - rhs2) <= 1e-6 * scale, (lhs2, rhs2, scale)
    # linearity in x
    out3 = torch.empty_like(out)
    L.call("icg_conv2d_fprop", (2.0 * x2).contiguous(memory_format=torch.channels_last), w, None, None, out3, None,
           None, 0, B, H, H, Cin, Cout, 3, 0, 1.0)
    assert rel_l2(out3, 2.0 * out2) < 1e-6


def test_bn_statistics_property_full_size():
    """After the fused apply, activations of a [64,96,256,256]-shaped tensor (batch 8 here) have zero mean /
    unit variance per channel: checked through the stand-alone apply at full spatial size."""
    import ic_gan_amd.ops as ops
    B, C, H = 8, 96, 256
    x = (torch.randn(B, C, H, H, device=DEV) * 3 + 1.5).contiguous(memory_format=torch.channels_last)
    bn = ops.BNOpt(torch.zeros(C, device=DEV), torch.ones(C, device=DEV), 1e-5, 0.1, True, 1.0, None)
    y = ops.norm_act(x, bn, torch.zeros(B, C, device=DEV), torch.zeros(B, C, device=DEV), relu=False)
    m = y.double().mean((0, 2, 3))
    v = y.double().var((0, 2, 3), unbiased=False)
    assert float(m.abs().max()) < 1e-4 and float((v - 1).abs().max()) < 1e-3
    ref_mean = x.double().mean((0, 2, 3))
    assert float((bn.running_mean.double() - 0.1 * ref_mean).abs().max()) < 1e-5


def test_full_size_step_cfg3_shape_is_finite():
    """One G+D step at cfg3's shape (256x256, ch=96, class+instance conditioning) with a reduced batch:
    finite losses, images in [-1, 1], every parameter receives a finite gradient / update."""
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    import ic_gan_amd.BigGAN as M
    cfg = dict(dim_z=120, shared_dim=128, shared_dim_feat=512, hier=True, n_classes=1000, SN_eps=1e-6, BN_eps=1e-5,
               G_ch=96, D_ch=96, G_attn="64", D_attn="64", resolution=256, class_cond=True, instance_cond=True,
               toggle_grads=True, num_D_steps=1, num_D_accumulations=1, num_G_accumulations=1, split_D=False,
               DiffAugment="", DA=False, D_ortho=0.0, G_ortho=0.0, ema=True)
    _, G, D, _, _ = _build(cfg)          # synthetic weights (orthogonal init of 1536x13824 matrices is slow on CPU)
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True}).to(DEV)
    ema = utils.ema(G, G_ema, 0.9999, 0)
    opt_d = FusedAdam(D.parameters(), lr=1e-4, betas=(0.0, 0.999), eps=1e-6)
    opt_g = FusedAdam(G.parameters(), lr=4e-5, betas=(0.0, 0.999), eps=1e-6)
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    gb = 4
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=1)
    train = train_fns.GAN_training_function(G, D, GD, ema, {"itr": 1}, cfg, samp, embedded_optimizers=False, device=DEV,
                                            batch_size=gb)
    x, y, f = synth.synth_batch(cfg, gb, seed=2)
    before = {n: p.detach().clone() for n, p in G.named_parameters()}
    m = train(_d(x), _d(y), _d(f))
    assert all(np.isfinite(v) for v in m.values()), m
    for n, p in list(G.named_parameters()) + list(D.named_parameters()):
        assert torch.isfinite(p).all(), n
    changed = sum(int(not torch.equal(p.detach(), before[n])) for n, p in G.named_parameters())
    assert changed >= 0.9 * len(before)
    with torch.no_grad():
        z, lab, fg = _cond(cfg, G.dim_z, gb, 5)
        img = G_ema(_d(z), _d(lab), _d(fg))
    assert img.shape == (gb, 3, 256, 256) and torch.isfinite(img).all() and float(img.abs().max()) <= 1.0

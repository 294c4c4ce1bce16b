"""GPU parity of the fused StyleGAN2 layer kernels (csrc/sg2_fused.hip): every C-ABI entry point against its CPU
restatement (oracle/kernel_ref.py) on the same seeded inputs, fp32 and fp16 storage, at toy shapes and at the layer shapes of
BASELINE.json configs[3] (64 ... 512 channels); then each fused layer (one autograd node) against the composed operator graph on the
GPU -- outputs and all first-order gradients."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as R

pytestmark = pytest.mark.gpu


def _L():
    import ic_gan_amd._lib as L
    return L


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, ref, tol, what):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what + ": non-finite"
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "%s: max err %.3e of max|ref| %.3e (%.2e rel), rel L2 %.2e" % (
        what, err, scale, err / scale, float((got - ref).norm() / (ref.norm() + 1e-30)))


def run_pair(name, args, outs):
    L = _L()
    dargs = [a.cuda() if isinstance(a, torch.Tensor) else a for a in args]
    L.call(name, *dargs)
    torch.cuda.synchronize()
    getattr(R, name)(*args)
    return [(dargs[i], args[i]) for i in outs]


def act(n, hw, c, half, seed, scale=1.0):
    t = rnd(n, hw, c, seed=seed, scale=scale)
    return t.half() if half else t


SHAPES = [(2, 64, 16), (3, 100, 32), (2, 37, 64), (2, 1024, 512), (2, 4096, 64), (4, 16, 512)]      # N, HW, C


@pytest.mark.parametrize("shape", [(24, 16, 3, 1, 0), (40, 24, 3, 1, 1), (16, 33, 1, 0, 0), (512, 512, 3, 1, 1), (64, 128, 3, 0, 0), (3, 64, 1, 0, 0)])
@pytest.mark.parametrize("half", [False, True])
def test_weight_prep(shape, half):
    from ic_gan_amd import ops
    O, I, Rk, prenorm, flip = shape
    dt = torch.float16 if half else torch.float32
    w = rnd(O, I, Rk, Rk, seed=1)
    gain = float(np.float32(1 / np.sqrt(I * Rk * Rk))) if prenorm else 0.37

    def bufs(dev):
        return dict(w=w.to(dev), w_fwd=torch.empty(O, Rk, Rk, I, dtype=dt, device=dev), w_adj=torch.empty(I, Rk, Rk, O, dtype=dt, device=dev),
                    wsq=torch.empty(O, I, device=dev), wscale=torch.empty(O, device=dev),
                    warg=torch.empty(O, dtype=torch.int32, device=dev) if prenorm else None, prenorm=prenorm, gain=gain, flip=flip)
    g, c = bufs("cuda"), bufs("cpu")
    ops.sg2_weight_prep_multi([g, dict(g, w_adj=None, wsq=None)])          # two layers in one call (the second rewrites the same outputs)
    torch.cuda.synchronize()
    R.sg2_weight_prep_ref([c])
    close(g["wscale"], c["wscale"], 1e-6, "wscale")
    close(g["wsq"], c["wsq"], 1e-5, "wsq")
    close(g["w_fwd"].float(), c["w_fwd"].float(), 1e-3 if half else 1e-6, "w_fwd")
    close(g["w_adj"].float(), c["w_adj"].float(), 1e-3 if half else 1e-6, "w_adj")
    if prenorm:
        assert torch.equal(g["warg"].cpu(), c["warg"])


@pytest.mark.parametrize("N,I,O,prenorm,demod", [(3, 16, 24, 0, 1), (4, 100, 33, 1, 1), (16, 512, 512, 1, 1), (2, 64, 0, 0, 0), (16, 512, 256, 0, 1)])
def test_style_prep(N, I, O, prenorm, demod):
    lin, bias = rnd(N, I, seed=1), rnd(I, seed=2) + 1
    wsq = rnd(max(O, 1), I, seed=3).square() if demod else None
    s, smax, sarg, d = torch.empty(N, I), torch.empty(N), torch.empty(N, dtype=torch.int32), torch.empty(N, max(O, 1))
    args = [lin, bias, 1.0, 0.7, wsq, N, I, O, prenorm, s, smax if prenorm else None, sarg if prenorm else None, d if demod else None]
    outs = [9] + ([10, 11] if prenorm else []) + ([12] if demod else [])
    for k, (g, r) in zip(outs, run_pair("icg_sg2_style_prep", args, outs)):
        if k == 11:
            assert torch.equal(g.cpu(), r)
        else:
            close(g, r, 2e-5, "style_prep out %d" % k)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("half", [False, True])
def test_modulate_and_its_gradient(shape, half):
    N, HW, C = shape
    dt = 1 if half else 0
    x, s = act(N, HW, C, half, 1), rnd(N, C, seed=2)
    xs = torch.empty_like(x)
    ((g, r),) = run_pair("icg_sg2_modulate", [x, s, xs, N, HW, C, dt], [2])
    close(g.float(), r.float(), 1e-3 if half else 1e-6, "modulate")
    dxs = act(N, HW, C, half, 3)
    dx, ds = torch.empty_like(x), torch.empty(N, C)
    nb = R.icg_sg2_rows_workspace_bytes(N, HW, C, C, dt)
    assert nb == _L().query("icg_sg2_rows_workspace_bytes", N, HW, C, C, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    (gdx, rdx), (gds, rds) = run_pair("icg_sg2_modulate_bwd", [dxs, x, s, dx, ds, N, HW, C, dt, ws, nb], [3, 4])
    close(gdx.float(), rdx.float(), 1e-3 if half else 1e-6, "modulate_bwd dx")
    close(gds, rds, 2e-5, "modulate_bwd ds")


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("variant", ["full", "per_sample_noise", "bias_act_only", "noise_only", "demod_only"])
def test_act_forward_and_backward(shape, half, variant):
    N, HW, O = shape
    dt = 1 if half else 0
    c = act(N, HW, O, half, 1)
    d = rnd(N, O, seed=2).abs() + 0.5 if variant in ("full", "per_sample_noise", "demod_only") else None
    noise = None
    bstride = 0
    if variant in ("full", "noise_only"):
        noise = rnd(HW, seed=3)
    if variant == "per_sample_noise":
        noise, bstride = rnd(N, HW, seed=3), HW
    strength = torch.tensor([0.3]) if noise is not None else None
    bias = rnd(O, seed=4) if variant != "demod_only" else None
    act_id, gain, clamp = (3, float(np.sqrt(2)), 1.5) if variant != "noise_only" else (1, 1.0, -1.0)
    y = torch.empty_like(c)
    ((g, r),) = run_pair("icg_sg2_act_fwd", [c, d, noise, bstride, strength, bias, y, N, HW, O, act_id, 0.2, gain, clamp, dt], [6])
    close(g.float(), r.float(), 2e-3 if half else 2e-6, "act_fwd")
    dy = act(N, HW, O, half, 5)
    dc, sums, tot = torch.empty_like(c), torch.empty(N, 2 * O + 1), torch.empty(2 * O + 1)
    nb = R.icg_sg2_rows_workspace_bytes(N, HW, O, 2 * O + 1, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    args = [dy, r, c if d is not None else None, d, noise, bstride, dc, sums, tot, N, HW, O, act_id, 0.2, gain, clamp, dt, ws, nb]
    (gdc, rdc), (gs, rs), (gt, rt) = run_pair("icg_sg2_act_bwd", args, [6, 7, 8])
    close(gdc.float(), rdc.float(), 2e-3 if half else 2e-6, "act_bwd dc")
    close(gs, rs, 3e-5, "act_bwd per-sample sums")
    close(gt, rt, 3e-5, "act_bwd batch sums")


@pytest.mark.parametrize("N,I,O,demod", [(3, 16, 24, 1), (4, 100, 33, 1), (16, 512, 512, 1), (2, 64, 0, 0)])
def test_style_backward(N, I, O, demod):
    ds = rnd(N, I + 5, seed=1)
    dd, d, s = rnd(N, 2 * max(O, 1) + 1, seed=2), rnd(N, max(O, 1), seed=3).abs() + 0.5, rnd(N, I, seed=4)
    wsq = rnd(max(O, 1), I, seed=5).square()
    nblk = (I + 63) // 64
    g, pdot, t = torch.empty(N, I), torch.empty(N, nblk), torch.empty(N, max(O, 1))
    args = [ds, I + 5, dd if demod else None, 2 * O + 1, d if demod else None, s, wsq if demod else None, N, I, O, g, pdot, t if demod else None]
    outs = [10, 11] + ([12] if demod else [])
    for k, (a, b) in zip(outs, run_pair("icg_sg2_style_bwd", args, outs)):
        close(a, b, 3e-5, "style_bwd out %d" % k)


@pytest.mark.parametrize("N,I,K,norm", [(3, 16, 24, 0), (4, 100, 33, 1), (16, 512, 512, 1), (16, 512, 2048, 0), (16, 1, 512, 0), (64, 300, 40, 0)])
def test_fc_backward(N, I, K, norm):
    g, x, W = rnd(N, I, seed=1), rnd(N, K, seed=2), rnd(I, K, seed=3)
    nblk = (I + 63) // 64
    smax = rnd(N, seed=4) + 0.1
    sarg = torch.randint(0, I, (N,), generator=torch.Generator().manual_seed(5), dtype=torch.int32)
    pdot = rnd(N, nblk, seed=6)
    dW, db, dx = torch.empty(I, K), torch.empty(I), torch.empty(N, K)
    args = [g, smax if norm else None, sarg if norm else None, pdot if norm else None, nblk if norm else 0, 0.6, x, W, N, I, K, 0.3, 1.7, dW, db, dx]
    for k, (a, b) in zip([13, 14, 15], run_pair("icg_sg2_fc_bwd", args, [13, 14, 15])):
        close(a, b, 3e-5, "fc_bwd out %d" % k)


@pytest.mark.parametrize("O,I,Rk,layout,demod,prenorm,round16", [(24, 16, 3, 0, 1, 0, 0), (40, 33, 3, 1, 1, 1, 1), (512, 512, 3, 0, 1, 1, 1), (64, 128, 3, 1, 0, 0, 0),
                                                                  (33, 16, 1, 0, 0, 0, 1), (3, 64, 1, 0, 1, 1, 0)])
def test_weight_backward(O, I, Rk, layout, demod, prenorm, round16):
    N, RR = 5, Rk * Rk
    dwc = rnd(RR, I, O, seed=1) if layout == 0 else rnd(RR, O, I, seed=1)
    t, s, w = rnd(N, O, seed=2), rnd(N, I, seed=3), rnd(O, I, Rk, Rk, seed=4)
    c0 = float(np.float32(1 / np.sqrt(I * RR)))
    m, arg = w.reshape(O, -1).abs().max(dim=1)
    wscale = (1.0 / m) * c0 if prenorm else torch.full((O,), 0.37)
    warg = arg.to(torch.int32)
    dw = torch.empty(O, I, Rk, Rk)
    nb = R.icg_sg2_weight_bwd_workspace_bytes(O, I)
    assert nb == _L().query("icg_sg2_weight_bwd_workspace_bytes", O, I)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    args = [dwc, layout, t if demod else None, s if demod else None, N if demod else 0, w, wscale, warg if prenorm else None, prenorm, c0, round16,
            dw, O, I, Rk, ws if prenorm else None, nb if prenorm else 0]
    ((a, b),) = run_pair("icg_sg2_weight_bwd", args, [11])
    close(a, b, 3e-5, "weight_bwd")


@pytest.mark.parametrize("N,HW,C", [(2, 64, 16), (3, 100, 32), (2, 4096, 64), (2, 1024, 512), (2, 256, 1024)])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("clamp,with_img", [(-1.0, False), (0.8, True)])
def test_torgb_forward_and_backward(N, HW, C, half, clamp, with_img):
    dt = 1 if half else 0
    if not R.icg_sg2_torgb_applies(C, dt):
        assert not _L().query("icg_sg2_torgb_applies", C, dt)
        pytest.skip("shape not served")
    assert _L().query("icg_sg2_torgb_applies", C, dt)
    x, s, w, bias = act(N, HW, C, half, 1), rnd(N, C, seed=2), rnd(3, C, seed=3, scale=0.1), rnd(3, seed=4)
    img = rnd(N, 3, HW, seed=5) if with_img else None
    out, y = torch.empty(N, 3, HW), torch.empty(N, HW, 3, dtype=x.dtype)
    (go, ro), (gy, ry) = run_pair("icg_sg2_torgb_fwd", [x, s, w, bias, clamp, img, out, y, N, HW, C, dt], [6, 7])
    close(go, ro, 2e-3 if half else 1e-5, "torgb img")
    close(gy.float(), ry.float(), 2e-3 if half else 1e-5, "torgb y")
    dimg = rnd(N, 3, HW, seed=6)
    dx, sums, tot = torch.empty_like(x), torch.empty(N, 4 * C + 3), torch.empty(4 * C + 3)
    nb = _L().query("icg_sg2_torgb_bwd_workspace_bytes", N, HW, C, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    args = [dimg, ry, x, s, w, clamp, 1, dx, sums, tot, N, HW, C, dt, ws, nb]
    (gdx, rdx), (gs, rs), (gt, rt) = run_pair("icg_sg2_torgb_bwd", args, [7, 8, 9])
    close(gdx.float(), rdx.float(), 2e-3 if half else 2e-6, "torgb_bwd dx")
    close(gs, rs, 5e-5, "torgb_bwd per-sample sums")
    close(gt, rt, 5e-5, "torgb_bwd batch sums")


# ------------------------------------------------------------------------------------------------ whole layers, fused vs composed
def _init(mod, seed):
    for i, (n, p) in enumerate(mod.named_parameters()):
        with torch.no_grad():
            v = rnd(*p.shape, seed=seed * 100 + i) if p.dim() else rnd(1, seed=seed * 100 + i)[0]
            if n.endswith("bias"):
                v = 0.3 * v + (1.0 if "affine" in n else 0.0)
            if n.endswith("noise_strength"):
                v = 0.3 * v
            p.copy_(v)
    for i, (n, b) in enumerate(mod.named_buffers()):          # `noise_const` is torch.randn at construction: seeded here like the parameters
        if b.is_floating_point() and "noise_const" in n:
            with torch.no_grad():
                b.copy_(rnd(*b.shape, seed=seed * 100 + 50 + i))


def _grads(fn, params, inputs, fused):
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    for p in params:
        p.grad = None
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    if fused:
        with FL.first_order():
            y = fn(*ins)
    else:
        y = fn(*ins)
    r = rnd(*y.shape, seed=99).to(y.device)
    (y.float() * r).sum().backward()
    return y.detach().float(), [t.grad.float() for t in ins], [p.grad.float() if p.grad is not None else None for p in params]


def _check(fn, mod, inputs, half):
    params, names = list(mod.parameters()), [n for n, _ in mod.named_parameters()]
    y0, gi0, gp0 = _grads(fn, params, inputs, False)
    y1, gi1, gp1 = _grads(fn, params, inputs, True)
    close(y1, y0, 2e-3 if half else 2e-5, "output")
    for i, (a, b) in enumerate(zip(gi1, gi0)):
        close(a, b, 6e-3 if half else 5e-5, "grad input %d" % i)
    for n, a, b in zip(names, gp1, gp0):
        assert (a is None) == (b is None), n
        if a is not None:
            # (a scalar noise strength is one sum of ~1e5 ... 1e6 signed fp16 products: the composed path rounds partial results to fp16)
            close(a, b, (3e-2 if a.numel() == 1 else 6e-3) if half else 5e-5, "grad " + n)


@pytest.mark.parametrize("cin,cout,res,up,half,noise_mode,n", [(512, 512, 16, 1, False, "random", 4), (512, 512, 16, 2, False, "const", 4),
                                                                (512, 512, 32, 2, True, "random", 4), (512, 256, 64, 2, True, "const", 2),
                                                                (128, 64, 64, 2, True, "random", 2), (64, 64, 64, 1, True, "const", 2),
                                                                (32, 32, 16, 1, False, "none", 3)])
def test_synthesis_layer_fused_equals_composed_hip(cin, cout, res, up, half, noise_mode, n, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.SynthesisLayer(cin, cout, w_dim=512, resolution=res, up=up, conv_clamp=256).cuda()
    _init(layer, 3)
    draws = rnd(n, 1, res, res, seed=77).cuda()
    monkeypatch.setattr(N, "_randn", lambda shape, device: draws.clone())
    x = rnd(n, cin, res // up, res // up, seed=5).cuda().contiguous(memory_format=torch.channels_last)
    w = rnd(n, 512, seed=6).cuda()
    if half:
        x = x.half()
    _check(lambda x, w: layer(x, w, noise_mode=noise_mode, fused_modconv=False), layer, [x, w], half)


@pytest.mark.parametrize("cin,res,half", [(512, 16, False), (64, 64, True), (256, 32, True)])
def test_torgb_layer_fused_equals_composed_hip(cin, res, half):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.ToRGBLayer(cin, 3, w_dim=512, conv_clamp=256).cuda()
    _init(layer, 4)
    x = rnd(2, cin, res, res, seed=5).cuda().contiguous(memory_format=torch.channels_last)
    if half:
        x = x.half()
    w, img = rnd(2, 512, seed=6).cuda(), rnd(2, 3, res, res, seed=8).cuda()
    _check(lambda x, w, img: layer(x, w, fused_modconv=False, img=img), layer, [x, w, img], half)


@pytest.mark.parametrize("cin,cout,k,down,act,bias,gain,half,res", [(64, 64, 3, 1, "lrelu", True, 1.0, True, 64), (64, 128, 3, 2, "lrelu", True, 0.7071, True, 64),
                                                                    (64, 128, 1, 2, "linear", False, 0.7071, True, 64), (512, 512, 3, 2, "lrelu", True, 0.7071, False, 16),
                                                                    (512, 512, 3, 1, "lrelu", True, 1.0, False, 8), (3, 64, 1, 1, "lrelu", True, 1.0, True, 64)])
def test_conv2d_layer_fused_equals_composed_hip(cin, cout, k, down, act, bias, gain, half, res):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.Conv2dLayer(cin, cout, kernel_size=k, bias=bias, activation=act, down=down, conv_clamp=256 if bias else None).cuda()
    _init(layer, 5)
    x = rnd(2, cin, res, res, seed=5).cuda().contiguous(memory_format=torch.channels_last)
    if half:
        x = x.half()
    _check(lambda x: layer(x, gain=gain), layer, [x], half)


def test_fully_connected_fused_equals_composed_hip():
    from ic_gan_amd.stylegan2 import networks as N
    for act, fin, fout in (("lrelu", 512, 512), ("linear", 2048, 512), ("linear", 512, 1)):
        layer = N.FullyConnectedLayer(fin, fout, activation=act, lr_multiplier=0.01 if act == "lrelu" else 1).cuda()
        _init(layer, 6)
        _check(lambda x: layer(x), layer, [rnd(16, fin, seed=5).cuda()], False)


# ------------------------------------------------------------------------------------------------ epilogues fused into the producers
@pytest.mark.parametrize("B,H,W,Cin,Cout,Rk,stride,pad", [(2, 16, 16, 64, 64, 3, 1, 1), (2, 20, 12, 32, 128, 3, 1, 1), (3, 17, 17, 64, 96, 3, 2, 0),
                                                           (2, 32, 32, 512, 512, 3, 1, 1), (2, 8, 8, 128, 64, 1, 1, 0)])
@pytest.mark.parametrize("variant", ["full", "per_sample_noise", "bias_act_only"])
def test_hconv_with_layer_epilogue(B, H, W, Cin, Cout, Rk, stride, pad, variant):
    Ho, Wo = (H + 2 * pad - Rk) // stride + 1, (W + 2 * pad - Rk) // stride + 1
    x = rnd(B, H, W, Cin, seed=1).half()
    w = rnd(Cout, Rk, Rk, Cin, seed=2, scale=1 / np.sqrt(Cin * Rk * Rk)).half()
    d = rnd(B, Cout, seed=3).abs() + 0.5 if variant != "bias_act_only" else None
    noise, nbs = (rnd(Ho * Wo, seed=4), 0) if variant == "full" else ((rnd(B, Ho * Wo, seed=4), Ho * Wo) if variant == "per_sample_noise" else (None, 0))
    strength = torch.tensor([0.3]) if noise is not None else None
    bias = rnd(Cout, seed=5)
    c, y = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16), torch.empty(B, Ho, Wo, Cout, dtype=torch.float16)
    args = [x, w, c if d is not None else None, y, d, noise, nbs, strength, bias, 3, 0.2, float(np.sqrt(2)), 1.5, B, H, W, Cin, Ho, Wo, Cout, Rk, stride, pad]
    outs = run_pair("icg_conv2d_g_fprop_f16_act", args, [3] + ([2] if d is not None else []))
    close(outs[0][0].float(), outs[0][1].float(), 3e-3, "y")           # (one fp16 ulp of c moves y by one ulp: fp32 vs fp64 accumulation)
    if d is not None:
        close(outs[1][0].float(), outs[1][1].float(), 2e-3, "c")
    # and bit-for-bit what the two separate kernels give on the GPU (where the stand-alone activation kernel takes the width)
    L = _L()
    if not L.query("icg_sg2_rows_applies", Cout, 1):
        return
    xg, wg = x.cuda(), w.cuda()
    c2, y2 = torch.empty_like(c, device="cuda"), torch.empty_like(y, device="cuda")
    L.call("icg_conv2d_g_fprop_f16", xg, wg, c2, B, H, W, Cin, Ho, Wo, Cout, Rk, stride, pad, 0)
    dv = lambda t: None if t is None else t.cuda()
    L.call("icg_sg2_act_fwd", c2, dv(d), dv(noise), nbs, dv(strength), dv(bias), y2, B, Ho * Wo, Cout, 3, 0.2, float(np.sqrt(2)), 1.5, 1)
    assert torch.equal(y2, outs[0][0]), "fused epilogue differs from conv + act_fwd: %d elements" % int((y2 != outs[0][0]).sum())


@pytest.mark.parametrize("N,H,W,C", [(2, 17, 17, 64), (3, 9, 13, 16), (2, 65, 65, 128), (2, 33, 33, 512)])
@pytest.mark.parametrize("half", [False, True])
def test_fir_with_layer_epilogue(N, H, W, C, half):
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    f = U.setup_filter([1, 3, 3, 1])
    oh, ow = H + 2 - 4 + 1, W + 2 - 4 + 1
    x = act(N, H * W, C, half, 1).reshape(N, H, W, C)
    d, noise, strength, bias = rnd(N, C, seed=2).abs() + 0.5, rnd(N, oh * ow, seed=3), torch.tensor([0.3]), rnd(C, seed=4)
    c, y = torch.empty(N, oh, ow, C, dtype=x.dtype), torch.empty(N, oh, ow, C, dtype=x.dtype)
    args = [x, f, c, y, d, noise, oh * ow, strength, bias, N, C, H, W, 4, 4, 1, 1, 1, 1, 0, 4.0, oh, ow, 3, 0.2, float(np.sqrt(2)), 2.0, 1 if half else 0]
    (gc, rc), (gy, ry) = run_pair("icg_sg2_fir_act_fwd", args, [2, 3])
    close(gc.float(), rc.float(), 2e-3 if half else 1e-5, "c")
    close(gy.float(), ry.float(), 3e-3 if half else 1e-5, "y")
    # the plain blur form (no epilogue) with a flipped asymmetric filter and asymmetric / negative padding, 4 x 4 and 3 x 5 taps
    for taps, pad in (([1, 2, 3, 4], (2, 1, 0, 3)), ([1, 3, 3, 1], (-1, 2, 1, -1))):
        f4 = U.setup_filter(taps)
        for ff in (f4, f4[:3, :].contiguous() if f4.dim() == 2 else f4):
            fh, fw = ff.shape
            o2h, o2w = H + pad[2] + pad[3] - fh + 1, W + pad[0] + pad[1] - fw + 1
            y2 = torch.empty(N, o2h, o2w, C, dtype=x.dtype)
            a2 = [x, ff, None, y2, None, None, 0, None, None, N, C, H, W, fh, fw, pad[0], pad[1], pad[2], pad[3], 1, 0.7, o2h, o2w, 1, 0.2, 1.0, -1.0,
                  1 if half else 0]
            ((g2, r2),) = run_pair("icg_sg2_fir_act_fwd", a2, [3])
            close(g2.float(), r2.float(), 2e-3 if half else 1e-5, "blur %dx%d" % (fh, fw))


@pytest.mark.parametrize("B,H,W,Cin,Cout,Rk,stride,pad,zins", [(2, 16, 16, 64, 64, 3, 1, 1, 0), (3, 12, 11, 32, 128, 3, 1, 1, 0), (2, 32, 32, 512, 512, 3, 1, 1, 0),
                                                                (3, 16, 16, 64, 128, 3, 1, 2, 2), (2, 32, 32, 256, 64, 3, 1, 2, 2), (2, 12, 12, 128, 64, 1, 1, 0, 0)])
@pytest.mark.parametrize("epilogue", [False, True])
def test_modulated_convolution_in_one_launch(B, H, W, Cin, Cout, Rk, stride, pad, zins, epilogue):
    """icg_modconv2d_f16: style scale on the A fragments (+ the layer epilogue) = bit for bit the separate kernels' result"""
    L = _L()
    if zins:
        Ho, Wo = (H - 1) * 2 + Rk - 2 * (Rk - 1 - pad), (W - 1) * 2 + Rk - 2 * (Rk - 1 - pad)
        Ho, Wo = 2 * H + 1, 2 * W + 1
        epilogue = False
    else:
        Ho, Wo = (H + 2 * pad - Rk) // stride + 1, (W + 2 * pad - Rk) // stride + 1
    assert L.query("icg_modconv2d_f16_applies", Cin, Cout, Rk, stride, zins, Ho, Wo) == R.icg_modconv2d_f16_applies(Cin, Cout, Rk, stride, zins, Ho, Wo) == 1
    x = rnd(B, H, W, Cin, seed=1).half().cuda()
    w = rnd(Cout, Rk, Rk, Cin, seed=2, scale=1 / np.sqrt(Cin * Rk * Rk)).half().cuda()
    s = rnd(B, Cin, seed=3).cuda()
    d, noise, strength, bias = (rnd(B, Cout, seed=4).abs() + 0.5).cuda(), rnd(B, Ho * Wo, seed=5).cuda(), torch.tensor([0.3]).cuda(), rnd(Cout, seed=6).cuda()
    c1, y1 = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16, device="cuda"), torch.empty(B, Ho, Wo, Cout, dtype=torch.float16, device="cuda")
    L.call("icg_modconv2d_f16", x, s, w, c1, y1 if epilogue else None, d, noise, Ho * Wo, strength, bias, 3, 0.2, 1.4, 2.0, B, H, W, Cin, Ho, Wo, Cout,
           Rk, stride, pad, zins)
    xs, c2, y2 = torch.empty_like(x), torch.empty_like(c1), torch.empty_like(y1)
    L.call("icg_sg2_modulate", x, s, xs, B, H * W, Cin, 1)
    L.call("icg_conv2d_g_fprop_f16", xs, w, c2, B, H, W, Cin, Ho, Wo, Cout, Rk, stride, pad, zins)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2), "modulated convolution differs from modulate + conv: %d elements" % int((c1 != c2).sum())
    if epilogue and L.query("icg_sg2_rows_applies", Cout, 1):
        L.call("icg_sg2_act_fwd", c2, d, noise, Ho * Wo, strength, bias, y2, B, Ho * Wo, Cout, 3, 0.2, 1.4, 2.0, 1)
        torch.cuda.synchronize()
        assert torch.equal(y1, y2)
    # and the CPU restatement
    cr = torch.empty(B, Ho, Wo, Cout, dtype=torch.float16)
    R.icg_modconv2d_f16(x.cpu(), s.cpu(), w.cpu(), cr, None, None, None, 0, None, None, 3, 0.2, 1.4, 2.0, B, H, W, Cin, Ho, Wo, Cout, Rk, stride, pad, zins)
    close(c1.float(), cr.float(), 2e-3, "c vs kernel_ref")


@pytest.mark.parametrize("N,HW,O", [(2, 64, 16), (3, 100, 32), (2, 4096, 64), (2, 777, 128), (2, 256, 512)])
@pytest.mark.parametrize("half", [False, True])
def test_fromrgb_forward_and_backward(N, HW, O, half):
    dt = 1 if half else 0
    if not R.icg_sg2_fromrgb_applies(O, dt):
        assert not _L().query("icg_sg2_fromrgb_applies", O, dt)
        pytest.skip("shape not served")
    cast = (lambda t: t.half()) if half else (lambda t: t)
    x, w, bias = cast(rnd(N, 3, HW, seed=1)), cast(rnd(O, 3, seed=2, scale=0.5)), rnd(O, seed=3)
    y = torch.empty(N, HW, O, dtype=x.dtype)
    ((gy, ry),) = run_pair("icg_sg2_fromrgb_fwd", [x, w, bias, y, N, HW, O, 3, 0.2, float(np.sqrt(2)), 1.2, dt], [3])
    close(gy.float(), ry.float(), 2e-3 if half else 2e-6, "fromrgb y")
    dy = act(N, HW, O, half, 5)
    dimg, tot = torch.empty_like(x), torch.empty(4 * O)
    nb = R.icg_sg2_rows_workspace_bytes(N, HW, O, 4 * O, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    (gd, rd), (gt, rt) = run_pair("icg_sg2_fromrgb_bwd", [dy, ry, x, w, dimg, tot, N, HW, O, 3, 0.2, float(np.sqrt(2)), 1.2, dt, ws, nb], [4, 5])
    close(gd.float(), rd.float(), 3e-3 if half else 1e-5, "fromrgb dimg")
    close(gt, rt, 5e-5, "fromrgb sums")


# ------------------------------------------------------------------------------------------------ second-order kernels and layers
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("half", [False, True])
def test_second_order_row_kernels(shape, half):
    N, HW, C = shape
    dt = 1 if half else 0
    x, g = act(N, HW, C, half, 1), act(N, HW, C, half, 2)
    a, b = rnd(N, C, seed=3), rnd(N, C, seed=4)
    u = torch.empty_like(x)
    for with_g in (True, False):
        ((gu, ru),) = run_pair("icg_sg2_mod2", [x, a, g if with_g else None, b if with_g else None, u, N, HW, C, dt], [4])
        close(gu.float(), ru.float(), 3e-3 if half else 2e-6, "mod2")
    dy, y, c, cdc = act(N, HW, C, half, 5), act(N, HW, C, half, 6), act(N, HW, C, half, 7), act(N, HW, C, half, 8)
    d, cdd = rnd(N, C, seed=9).abs() + 0.5, rnd(N, C, seed=10)
    cdy, cc, sums = torch.empty_like(x), torch.empty_like(x), torch.empty(N, C)
    nb = R.icg_sg2_rows_workspace_bytes(N, HW, C, C, dt)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    args = [dy, y, c, cdc, d, cdd, cdy, cc, sums, N, HW, C, 3, 0.2, float(np.sqrt(2)), 1.0, dt, ws, nb]
    (g1, r1), (g2, r2), (g3, r3) = run_pair("icg_sg2_act_bwd2", args, [6, 7, 8])
    close(g1.float(), r1.float(), 3e-3 if half else 2e-6, "act_bwd2 cdy")
    close(g2.float(), r2.float(), 3e-3 if half else 2e-6, "act_bwd2 cc")
    close(g3, r3, 5e-5, "act_bwd2 sums")


def test_weight_backward_with_general_table_cotangent():
    O, I, Rk, N = 40, 33, 3, 5
    RR = Rk * Rk
    dwc, Q, w = rnd(RR, I, O, seed=1), rnd(O, I, seed=2), rnd(O, I, Rk, Rk, seed=4)
    c0 = float(np.float32(1 / np.sqrt(I * RR)))
    m, arg = w.reshape(O, -1).abs().max(dim=1)
    wscale, warg = (1.0 / m) * c0, arg.to(torch.int32)
    dw = torch.empty(O, I, Rk, Rk)
    nb = R.icg_sg2_weight_bwd_workspace_bytes(O, I)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    ((a, b),) = run_pair("icg_sg2_weight_bwd_q", [dwc, 0, None, None, 0, Q, w, wscale, warg, 1, c0, 0, dw, O, I, Rk, ws, nb], [12])
    close(a, b, 3e-5, "weight_bwd_q")


@pytest.mark.parametrize("N,HW,C", [(2, 64, 16), (3, 100, 32), (2, 4096, 64), (2, 1024, 512)])
@pytest.mark.parametrize("half", [False, True])
def test_torgb_second_order_kernel(N, HW, C, half):
    dt = 1 if half else 0
    x, cdx = act(N, HW, C, half, 1), act(N, HW, C, half, 7)
    s, w, a = rnd(N, C, seed=2), rnd(3, C, seed=3, scale=0.1), rnd(N, C, seed=8)
    dimg, cim = rnd(N, 3, HW, seed=6), rnd(N, 3, HW, seed=9)
    y = (rnd(N, HW, 3, seed=5) * 0.6)
    y = y.half() if half else y
    for with_cdx in (True, False):
        cdimg, cx, sums, tot = torch.empty(N, 3, HW), torch.empty_like(x), torch.empty(N, 4 * C), torch.empty(4 * C)
        nb = _L().query("icg_sg2_torgb_bwd_workspace_bytes", N, HW, C, dt)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8)
        args = [dimg, y, x, s, w, a, cdx if with_cdx else None, cim if with_cdx else None, 0.8, 1, cdimg, cx, sums, tot, N, HW, C, dt, ws, nb]
        (g1, r1), (g2, r2), (g3, r3), (g4, r4) = run_pair("icg_sg2_torgb_bwd2", args, [10, 11, 12, 13])
        close(g1, r1, 3e-3 if half else 2e-5, "torgb_bwd2 cdimg")
        close(g2.float(), r2.float(), 3e-3 if half else 2e-6, "torgb_bwd2 cx")
        close(g3, r3, 5e-5, "torgb_bwd2 sums")
        close(g4, r4, 5e-5, "torgb_bwd2 tot")


def _second_order(fn, params, inputs, fused):
    import contextlib
    from ic_gan_amd.stylegan_ops import conv2d_gradfix, fused_layers as FL
    for p in params:
        p.grad = None
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    with (FL.second_order() if fused else contextlib.nullcontext()):
        y = fn(*ins)
        rs = rnd(*y.shape, seed=95).to(y.device).requires_grad_(True)
        ins.append(rs)
        # (small probes: at these widths the composed path's fp16 partial sums of the double backward overflow for O(1) cotangents)
        r = rnd(*y.shape, seed=99, scale=0.02 if y.dtype == torch.float16 else 1.0).to(y.device)
        with conv2d_gradfix.no_weight_gradients():
            g, gx = torch.autograd.grad([(y.float() * (r * rs)).sum()], [ins[1], ins[0]], create_graph=True, only_inputs=True)
        q, qx, qy = rnd(*g.shape, seed=98).to(y.device), rnd(*gx.shape, seed=97).to(y.device), rnd(*y.shape, seed=96).to(y.device)
        (g.square().sum() * 0.5 + (g * q).sum() + (gx.float() * qx).sum() + (y.float() * qy).sum() * 0.1).backward()
    return g.detach().float(), [t.grad.float() for t in ins], [p.grad.float() if p.grad is not None else None for p in params]


def _check2(fn, mod, inputs, half):
    params, names = list(mod.parameters()), [n for n, _ in mod.named_parameters()]
    g0, gi0, gp0 = _second_order(fn, params, inputs, False)
    g1, gi1, gp1 = _second_order(fn, params, inputs, True)
    close(g1, g0, 6e-3 if half else 5e-5, "path-length vector")
    for i, (a, b) in enumerate(zip(gi1, gi0)):
        close(a, b, 3e-2 if half else 2e-4, "second-order grad input %d" % i)
    for n, a, b in zip(names, gp1, gp0):
        if n.endswith("noise_strength") or (n.endswith("bias") and "affine" not in n) or a is None or b is None:
            continue
        close(a, b, 3e-2 if half else 2e-4, "second-order grad " + n)


@pytest.mark.parametrize("cin,cout,res,up,half,noise_mode,n", [(512, 512, 16, 1, False, "random", 4), (512, 512, 16, 2, False, "const", 4),
                                                                (512, 512, 32, 2, True, "random", 4), (128, 64, 64, 2, True, "random", 2),
                                                                (64, 64, 64, 1, True, "const", 2)])
def test_synthesis_layer_second_order_fused_equals_composed_hip(cin, cout, res, up, half, noise_mode, n, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.SynthesisLayer(cin, cout, w_dim=512, resolution=res, up=up, conv_clamp=256).cuda()
    _init(layer, 3)
    draws = rnd(n, 1, res, res, seed=77).cuda()
    monkeypatch.setattr(N, "_randn", lambda shape, device: draws.clone())
    x = rnd(n, cin, res // up, res // up, seed=5).cuda()            # (NCHW order: re-laid-out inside the node)
    w = rnd(n, 3, 512, seed=6).cuda()[:, 1]                         # (a non-contiguous slice, as ws.unbind gives)
    if half:
        x = x.half()
    _check2(lambda x, w: layer(x, w, noise_mode=noise_mode, fused_modconv=False), layer, [x, w], half)


@pytest.mark.parametrize("cin,res,half", [(512, 16, False), (64, 64, True)])
def test_torgb_layer_second_order_fused_equals_composed_hip(cin, res, half):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.ToRGBLayer(cin, 3, w_dim=512, conv_clamp=256).cuda()
    _init(layer, 4)
    x = rnd(2, cin, res, res, seed=5).cuda()
    if half:
        x = x.half()
    w, img = rnd(2, 512, seed=6).cuda(), rnd(2, 3, res, res, seed=8).cuda()
    _check2(lambda x, w, img: layer(x, w, fused_modconv=False, img=img), layer, [x, w, img], half)

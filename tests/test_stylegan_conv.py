"""conv2d_gradfix / conv2d_resample / modulated_conv2d against outputs and first- and second-order
gradients of the reference itself (tests/golden/stylegan_conv.npz, made by make_golden_stylegan_conv.py).

fp32 tolerance: 2e-4 of the tensor's rms for outputs and first-order gradients, 1e-3 for second-order ones (two chained
fp32 contractions in a different summation order than the CPU reference); measured errors are ~1e-6 / ~1e-5.
CPU variants route the C-ABI to oracle/kernel_ref.py (host logic: geometry algebra, adjoint weights, autograd wiring);
GPU variants run the HIP kernels."""
import os

import numpy as np
import pytest
import torch

from oracle import kernel_ref
from tests.helpers import GOLDEN_DIR
from tests.stylegan_cases import CONV, MODCONV, RESAMPLE, rnd

GOLD = np.load(os.path.join(GOLDEN_DIR, "stylegan_conv.npz"))
TOL1, TOL2 = 2e-4, 1e-3


@pytest.fixture
def emu(monkeypatch):
    kernel_ref.install(monkeypatch)


def _close(got, key, tol):
    ref = GOLD[key]
    got = got.detach().cpu().numpy()
    assert got.shape == ref.shape, (key, got.shape, ref.shape)
    scale = max(float(np.sqrt((ref.astype(np.float64) ** 2).mean())), 1e-6)
    err = float(np.abs(got - ref).max())
    assert err <= tol * scale + 1e-6, "%s: max abs err %.3e vs rms %.3e (tol %.1e)" % (key, err, scale, tol)


def _second_order(key, y, inputs, names, seed, dev):
    dy = rnd(tuple(y.shape), seed).to(dev).requires_grad_(True)
    g1 = torch.autograd.grad(y, inputs, dy, create_graph=True)
    for n, g in zip(names, g1):
        _close(g, f"{key}d{n}", TOL1)
    scalar = sum((g * rnd(tuple(g.shape), seed + 1 + i).to(dev)).sum() for i, g in enumerate(g1))
    g2 = torch.autograd.grad(scalar, list(inputs) + [dy], allow_unused=True)
    for n, g, ref in zip(list(names) + ["y"], g2, list(inputs) + [dy]):
        _close(g if g is not None else torch.zeros_like(ref), f"{key}dd{n}", TOL2)


def _conv_case(i, dev):
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    n, ci, h, w, co, r, s, p, tr, op = CONV[i]
    x = rnd((n, ci, h, w), 100 + i).to(dev).requires_grad_(True)
    wt = (rnd((ci, co, r, r) if tr else (co, ci, r, r), 200 + i) * (ci * r * r) ** -0.5).to(dev).requires_grad_(True)
    if tr:
        y = cg.conv_transpose2d(x, wt, stride=s, padding=p, output_padding=op)
    else:
        y = cg.conv2d(x, wt, stride=s, padding=p)
    _close(y, f"conv/{i}/y", TOL1)
    _second_order(f"conv/{i}/", y, (x, wt), ("x", "w"), 300 + 10 * i, dev)


def _resample_case(i, dev):
    from ic_gan_amd.stylegan_ops import conv2d_resample as cr, upfirdn2d as up_
    n, ci, h, w, co, k, up, down, pad, fw, ff, taps = RESAMPLE[i]
    x = rnd((n, ci, h, w), 400 + i).to(dev).requires_grad_(True)
    wt = (rnd((co, ci, k, k), 500 + i) * (ci * k * k) ** -0.5).to(dev).requires_grad_(True)
    f = up_.setup_filter(taps, device=dev) if taps is not None else None
    y = cr.conv2d_resample(x, wt, f=f, up=up, down=down, padding=pad, flip_weight=fw, flip_filter=ff)
    _close(y, f"rs/{i}/y", TOL1)
    _second_order(f"rs/{i}/", y, (x, wt), ("x", "w"), 600 + 10 * i, dev)


def _modconv_case(i, dev):
    from ic_gan_amd.stylegan_ops import modulated_conv2d, upfirdn2d as up_
    n, ci, h, w, co, k, up, demod, noise, fused = MODCONV[i]
    x = rnd((n, ci, h, w), 700 + i).to(dev).requires_grad_(True)
    wt = rnd((co, ci, k, k), 800 + i).to(dev).requires_grad_(True)
    st = (rnd((n, ci), 900 + i) * 0.5 + 1.0).to(dev).requires_grad_(True)
    nz = (rnd((n, 1, h * up, w * up), 950 + i) * 0.1).to(dev) if noise else None
    f = up_.setup_filter([1, 3, 3, 1], device=dev)
    y = modulated_conv2d(x=x, weight=wt, styles=st, noise=nz, up=up, padding=k // 2, resample_filter=f,
                         demodulate=demod, flip_weight=(up == 1), fused_modconv=fused)
    _close(y, f"mc/{i}/y", TOL1)
    _second_order(f"mc/{i}/", y, (x, wt, st), ("x", "w", "s"), 1000 + 10 * i, dev)


@pytest.mark.parametrize("i", range(len(CONV)))
def test_conv2d_gradfix_host_logic(i, emu):
    _conv_case(i, "cpu")


@pytest.mark.parametrize("i", range(len(RESAMPLE)))
def test_conv2d_resample_host_logic(i, emu):
    _resample_case(i, "cpu")


@pytest.mark.parametrize("i", range(len(MODCONV)))
def test_modulated_conv2d_host_logic(i, emu):
    _modconv_case(i, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(CONV)))
def test_conv2d_gradfix_hip(i):
    _conv_case(i, "cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(RESAMPLE)))
def test_conv2d_resample_hip(i):
    _resample_case(i, "cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(MODCONV)))
def test_modulated_conv2d_hip(i):
    _modconv_case(i, "cuda:0")


def test_no_weight_gradients_context(emu):
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    x = rnd((1, 4, 6, 6), 1).requires_grad_(True)
    w = rnd((4, 4, 3, 3), 2).requires_grad_(True)
    with cg.no_weight_gradients():
        y = cg.conv2d(x, w, padding=1)
        gx, gw = torch.autograd.grad(y.sum(), (x, w), allow_unused=True)
    assert gx is not None and gw is None
    assert cg.weight_gradients_disabled is False
    with pytest.raises(ValueError):
        cg.conv2d(x, w, padding=1, groups=3)
    with pytest.raises(NotImplementedError):
        cg.conv2d(x, w, padding=1, dilation=2)


@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_grouped_convolution_equals_torch(dev, transpose, monkeypatch):
    """conv2d / conv_transpose2d with groups > 1 (conv2d_gradfix.py:43-99 hands `groups` to cuDNN; here one groups=1 call per group):
    values, first- and second-order gradients against F.conv2d / F.conv_transpose2d on the same fp32 tensors."""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    if dev == "cpu":
        kernel_ref.install(monkeypatch)
    G, cin_g, cout_g = 3, 4, 8
    x = rnd((2, G * cin_g, 6, 6), 11).to(dev).requires_grad_(True)
    shape = (G * cin_g, cout_g, 3, 3) if transpose else (G * cout_g, cin_g, 3, 3)
    w = (rnd(shape, 12) * 0.2).to(dev).requires_grad_(True)
    b = rnd((G * cout_g,), 13).to(dev).requires_grad_(True)
    xr, wr, br = (t.detach().cpu().clone().requires_grad_(True) for t in (x, w, b))
    if transpose:
        y = cg.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1, groups=G)
        ref = torch.nn.functional.conv_transpose2d(xr, wr, br, stride=2, padding=1, output_padding=1, groups=G)
    else:
        y = cg.conv2d(x, w, b, stride=1, padding=1, groups=G)
        ref = torch.nn.functional.conv2d(xr, wr, br, stride=1, padding=1, groups=G)
    assert y.shape == ref.shape
    dy = rnd(tuple(ref.shape), 14)
    g = torch.autograd.grad(y, (x, w, b), dy.to(dev), create_graph=True)
    gr = torch.autograd.grad(ref, (xr, wr, br), dy, create_graph=True)
    p, q = rnd(tuple(x.shape), 15), rnd(tuple(w.shape), 16)
    g2 = torch.autograd.grad((g[0] * p.to(dev)).sum() + (g[1] * q.to(dev)).sum(), (x, w))
    g2r = torch.autograd.grad((gr[0] * p).sum() + (gr[1] * q).sum(), (xr, wr))
    for name, u, v in zip(("y", "dx", "dw", "db", "ddx", "ddw"), (y,) + tuple(g) + tuple(g2), (ref,) + tuple(gr) + tuple(g2r)):
        err = float((u.detach().cpu() - v.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(v.detach().abs().max())), "%s: %.3e" % (name, err)


@pytest.mark.parametrize("i", [k for k, c in enumerate(MODCONV) if c[-1]])
@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_fused_modconv_as_the_reference_grouped_convolution(dev, i, monkeypatch):
    """`fused_modconv=True` in the reference's literal form (per-sample weights, groups = batch: networks.py:64-73,100-117) against
    the goldens the reference produced on exactly that path, outputs and gradients of both orders -- and against the default
    route (activations scaled before and after one convolution) on the same inputs."""
    from ic_gan_amd.stylegan_ops import modconv, modulated_conv2d, upfirdn2d as up_
    if dev == "cpu":
        kernel_ref.install(monkeypatch)
    monkeypatch.setattr(modconv, "GROUPED_FUSED_MODCONV", True)
    _modconv_case(i, dev)
    n, ci, h, w, co, k, up, demod, noise, fused = MODCONV[i]
    x = rnd((n, ci, h, w), 700 + i).to(dev)
    wt = rnd((co, ci, k, k), 800 + i).to(dev)
    st = (rnd((n, ci), 900 + i) * 0.5 + 1.0).to(dev)
    nz = (rnd((n, 1, h * up, w * up), 950 + i) * 0.1).to(dev) if noise else None
    f = up_.setup_filter([1, 3, 3, 1], device=dev)
    kw = dict(x=x, weight=wt, styles=st, noise=nz, up=up, padding=k // 2, resample_filter=f, demodulate=demod, flip_weight=(up == 1),
              fused_modconv=True)
    grouped = modulated_conv2d(**kw)
    monkeypatch.setattr(modconv, "GROUPED_FUSED_MODCONV", False)
    plain = modulated_conv2d(**kw)
    assert float((grouped - plain).abs().max()) <= 2e-5 * float(plain.abs().max())


@pytest.mark.parametrize("m", [2, 4])
@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_conv2d_gradfix_winograd_path(dev, m, monkeypatch):
    """the wide-layer Winograd route of conv2d_gradfix (forced onto narrow layers here) reproduces the same reference
    goldens, including second-order gradients (the data gradient of a 'same' 3x3 conv is again a 'same' 3x3 conv)."""
    import ic_gan_amd.ops as _ops
    if dev == "cpu":
        kernel_ref.install(monkeypatch)
    monkeypatch.setattr(_ops, "WINOGRAD_MIN_CHANNELS", 4)
    monkeypatch.setattr(_ops, "WINOGRAD2_MIN_CHANNELS", 4)
    monkeypatch.setattr(_ops, "WINOGRAD4_MIN_CHANNELS", 4 if m == 4 else 10 ** 9)
    monkeypatch.setattr(_ops, "WINOGRAD4_WGRAD_MIN_CHANNELS", 4 if m == 4 else 10 ** 9)
    _conv_case(0, dev)        # 8 -> 12 channels, 9x9: odd size -> stays on the direct path
    _conv_case(9, dev)        # 32 -> 32 channels, 16x16, stride 1, pad 1 -> Winograd
    _resample_case(4, dev)    # plain 3x3 through conv2d_resample
    _modconv_case(0, dev)     # modulated conv, 8 -> 6 channels... (Cout % 4 != 0 -> direct)
    _modconv_case(3, dev)     # 8 -> 8 channels at 4x4 -> Winograd


@pytest.mark.parametrize("kind", ["same", "down", "tr"])
@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
def test_conv2d_gradfix_fp16_mfma_route(dev, kind, monkeypatch):
    """conv2d_gradfix on fp16 tensors (the reference's `num_fp16_res` blocks): the fp16-input MFMA route (icg_conv2d_g_fprop_f16) against
    the exact-fp32 kernels between two casts (FP16_MFMA = False) -- outputs, first- and second-order gradients of a 3x3 'same', a
    stride-2 and a transposed stride-2 layer.  Both routes round every tensor to fp16 at the same places; they differ by the
    accumulation order inside fp32, i.e. by an occasional fp16 rounding flip: 4 fp16 ulps of the value + 2e-3 of the tensor maximum."""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    if dev == "cpu":
        kernel_ref.install(monkeypatch)

    def run(route):
        monkeypatch.setattr(cg, "FP16_MFMA", route)
        x = rnd((4, 64, 16, 16), 3).to(dev).half().requires_grad_(True)
        shape = (64, 128, 3, 3) if kind == "tr" else (128, 64, 3, 3)
        wt = (rnd(shape, 4) * (64 * 9) ** -0.5).to(dev).half().requires_grad_(True)
        if kind == "same":
            y = cg.conv2d(x, wt, padding=1)
        elif kind == "down":
            y = cg.conv2d(x, wt, stride=2, padding=1)
        else:
            y = cg.conv_transpose2d(x, wt, stride=2, padding=0)
        assert y.dtype == torch.float16
        dy = rnd(tuple(y.shape), 5).to(dev).half().requires_grad_(True)
        gx, gw = torch.autograd.grad(y, (x, wt), dy, create_graph=True)
        assert gx.dtype == torch.float16 and gw.dtype == torch.float16
        s = (gx.float() * rnd(tuple(gx.shape), 6).to(dev)).sum() + (gw.float() * rnd(tuple(gw.shape), 7).to(dev)).sum()
        g2 = torch.autograd.grad(s, (x, wt, dy))
        return [t.detach().float().cpu() for t in (y, gx, gw) + tuple(g2)]

    a, b = run(True), run(False)
    for name, u, v in zip(("y", "dx", "dw", "ddx", "ddw", "ddy"), a, b):
        assert u.shape == v.shape and torch.isfinite(u).all()
        tol = 2e-3 * float(v.abs().max()) + 4e-3 * v.abs()
        bad = (u - v).abs() > tol
        assert not bad.any(), "%s %s: %d / %d beyond tolerance, worst %.3e" % (kind, name, int(bad.sum()), bad.numel(),
                                                                               float(((u - v).abs() - tol).max()))


def test_odd_channel_count_is_padded_onto_the_vector_path(emu):
    """conv2d with Cin % 4 != 0 (the 513-channel layer behind MinibatchStd, training/networks.py:706-712) pads x and w with zero
    channels; outputs and both gradients equal F.conv2d's, the padding never shows in the gradient shapes."""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    x = rnd((2, 9, 6, 6), 1).requires_grad_(True)
    w = (rnd((8, 9, 3, 3), 2) * 0.1).requires_grad_(True)
    y = cg.conv2d(x, w, padding=1)
    ref = torch.nn.functional.conv2d(x, w, padding=1)
    assert float((y - ref).detach().abs().max()) < 1e-5
    gx, gw = torch.autograd.grad((y * rnd(tuple(y.shape), 3)).sum(), (x, w))
    rx, rw = torch.autograd.grad((ref * rnd(tuple(y.shape), 3)).sum(), (x, w))
    assert gx.shape == x.shape and gw.shape == w.shape
    assert float((gx - rx).abs().max()) < 1e-5 and float((gw - rw).abs().max()) < 1e-5


@pytest.mark.parametrize("m", [4, 16, 40])
def test_linear_nt_all_orders(m, emu):
    """linear_nt (conv2d_gradfix._Matmul): x w^T with first- and second-order gradients against torch; m <= 32 rows takes the transposed-
    weight route in its data gradient (mode 1 -> mode 0 of the row-streaming GEMM)."""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    x = rnd((m, 160), 4).requires_grad_(True)
    w = (rnd((24, 160), 5) * 0.1).requires_grad_(True)
    x2, w2 = x.detach().clone().requires_grad_(True), w.detach().clone().requires_grad_(True)
    y, ref = cg.linear_nt(x, w), x2 @ w2.t()
    assert float((y - ref).detach().abs().max()) < 1e-5
    dy = rnd((m, 24), 6)
    g = torch.autograd.grad(y, (x, w), dy, create_graph=True)
    r = torch.autograd.grad(ref, (x2, w2), dy, create_graph=True)
    for a, b in zip(g, r):
        assert float((a - b).detach().abs().max()) < 1e-5
    s = (g[0] * rnd((m, 160), 7)).sum() + (g[1] * rnd((24, 160), 8)).sum()
    s2 = (r[0] * rnd((m, 160), 7)).sum() + (r[1] * rnd((24, 160), 8)).sum()
    for a, b in zip(torch.autograd.grad(s, (x, w)), torch.autograd.grad(s2, (x2, w2))):
        assert float((a - b).abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape", [(6, 3, 3, 4), (5, 1, 1, 7), (4, 4, 4, 8)])
def test_adjoint_weight_is_transpose_and_tap_flip(shape, dtype):
    """conv2d_gradfix._adjoint_weight (one gather along the reversed flattened tap axis) = permute + flip of both tap axes, values and
    the gradient that flows back through it (second-order terms of the data gradient)."""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix as cg
    w = rnd(shape, 21).to(dtype).requires_grad_(True)
    got = cg._adjoint_weight(w)
    want = w.permute(3, 1, 2, 0).flip(1, 2).contiguous()
    assert got.shape == want.shape and got.is_contiguous() and torch.equal(got, want)
    p = rnd(tuple(want.shape), 22).to(dtype)
    (ga,) = torch.autograd.grad((got * p).sum(), w)
    (gb,) = torch.autograd.grad((want * p).sum(), w)
    assert torch.equal(ga, gb)

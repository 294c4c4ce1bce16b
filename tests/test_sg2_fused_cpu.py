"""Host logic of the fused StyleGAN2 layers (ic_gan_amd/stylegan_ops/fused_layers.py) on CPU, kernels emulated by
oracle/kernel_ref.py: each fused layer -- ONE autograd node -- must give the outputs and ALL first-order gradients of the composed
operator graph it replaces (modconv.py / conv2d_resample.py / bias_act.py, which the reference-generated goldens of
tests/test_stylegan2.py pin), for every layer shape of the networks: fp32 and fp16 storage, up-sampling, down-sampling, noise
modes, clamps, gains.  fp32: equal to fp32 round-off; fp16: same rounding points, so equal up to a few fp16 ulps (the composed path
rounds its reductions to fp16, the fused kernels keep them in fp32)."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref


@pytest.fixture
def emu(monkeypatch):
    kernel_ref.install(monkeypatch)


def _rnd(shape, seed, scale=1.0):
    return torch.tensor(np.asarray(np.random.RandomState(seed).standard_normal(shape), dtype=np.float64) * scale, dtype=torch.float32).reshape(shape)


def _init(mod, seed):
    for i, (n, p) in enumerate(mod.named_parameters()):
        with torch.no_grad():
            v = _rnd(tuple(p.shape), seed * 100 + i)
            if n.endswith("bias"):
                v = 0.3 * v + (1.0 if "affine" in n else 0.0)
            if n.endswith("noise_strength"):
                v = 0.3 * v
            p.copy_(v)


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
    r = _rnd(tuple(y.shape), 99).to(y.dtype)
    (y.float() * r.float()).sum().backward()
    return y.detach().float(), [t.grad.float() for t in ins], [p.grad.float() if p.grad is not None else None for p in params]


def _cmp(a, b, tol, what):
    if a is None or b is None:
        assert a is None and b is None, what
        return
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max())
    assert a.shape == b.shape and err <= tol * scale, "%s: err %.3e of max %.3e (%.2e rel)" % (what, err, scale, err / scale)


def _check(fn, mod, inputs, half, monkeypatch, counts=None):
    from ic_gan_amd import _lib as L
    params = list(mod.parameters())
    names = [n for n, _ in mod.named_parameters()]
    seen = []
    orig = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (seen.append(name), orig(name, *a))[1])
    y0, gi0, gp0 = _grads(fn, params, inputs, fused=False)
    assert not any(n.startswith("icg_sg2_") for n in seen), "the composed path must not touch the fused kernels"
    del seen[:]
    y1, gi1, gp1 = _grads(fn, params, inputs, fused=True)
    assert any(n.startswith("icg_sg2_") for n in seen), "fused path not taken: %s" % sorted(set(seen))
    if counts is not None:
        counts.append(len(seen))
    tol = 4e-3 if half else 2e-5
    _cmp(y1, y0, 2e-3 if half else 1e-5, "output")
    for i, (a, b) in enumerate(zip(gi1, gi0)):
        _cmp(a, b, tol, "grad input %d" % i)
    for n, a, b in zip(names, gp1, gp0):
        _cmp(a, b, tol, "grad " + n)


SYN = [  # Cin, Cout, res(out), up, half, noise_mode, clamp, gain, N
    (16, 16, 8, 1, False, "const", None, 1.0, 3),
    (16, 32, 8, 2, False, "const", 256, 1.0, 2),
    (32, 16, 8, 1, True, "const", 256, 1.0, 3),
    (32, 32, 16, 2, True, "random", 256, 1.0, 2),
    (64, 32, 8, 2, True, "none", 2.0, float(np.sqrt(0.5)), 2),       # clamp active, resnet gain
    (8, 8, 4, 1, False, "random", 0.5, 1.0, 4),
    (64, 64, 16, 1, True, "const", 256, 1.0, 2),          # widths the fp16 MFMA kernel takes: modulation + epilogue inside the convolution
    (64, 64, 32, 2, True, "random", 256, 1.0, 2),         # ... up-sampling: modulation inside the transposed convolution, epilogue on the blur
]


@pytest.mark.parametrize("case", SYN)
def test_synthesis_layer_fused_equals_composed(case, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    cin, cout, res, up, half, noise_mode, clamp, gain, n = case
    layer = N.SynthesisLayer(cin, cout, w_dim=24, resolution=res, up=up, conv_clamp=clamp)
    _init(layer, 3)
    draws = _rnd((n, 1, res, res), 77)
    monkeypatch.setattr(N, "_randn", lambda shape, device: draws.clone())
    x = _rnd((n, cin, res // up, res // up), 5).contiguous(memory_format=torch.channels_last)
    w = _rnd((n, 24), 6)
    if half:
        x = x.half()
    _check(lambda x, w: layer(x, w, noise_mode=noise_mode, fused_modconv=False, gain=gain), layer, [x, w], half, monkeypatch)


@pytest.mark.parametrize("half,clamp,with_img,cin", [(False, None, False, 16), (True, 256, True, 32), (True, 0.7, True, 64), (False, 0.5, True, 512)])
def test_torgb_layer_fused_equals_composed(half, clamp, with_img, cin, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.ToRGBLayer(cin, 3, w_dim=24, conv_clamp=clamp)
    _init(layer, 4)
    n, res = 2, 8
    x = _rnd((n, cin, res, res), 5).contiguous(memory_format=torch.channels_last)
    w = _rnd((n, 24), 6)
    img = _rnd((n, 3, res, res), 8)
    if half:
        x = x.half()
    if with_img:
        _check(lambda x, w, img: layer(x, w, fused_modconv=False, img=img), layer, [x, w, img], half, monkeypatch)
    else:
        _check(lambda x, w: layer(x, w, fused_modconv=False), layer, [x, w], half, monkeypatch)


CONV = [  # Cin, Cout, k, down, act, bias, clamp, gain, half, res
    (16, 32, 3, 1, "lrelu", True, 256, 1.0, False, 8),
    (16, 32, 3, 2, "lrelu", True, 256, float(np.sqrt(0.5)), False, 8),
    (16, 32, 1, 2, "linear", False, None, float(np.sqrt(0.5)), False, 8),        # resnet skip: the gain rides in the weight
    (32, 32, 3, 1, "lrelu", True, 256, 1.0, True, 8),
    (32, 64, 3, 2, "lrelu", True, 1.5, float(np.sqrt(0.5)), True, 16),
    (32, 64, 1, 2, "linear", False, None, float(np.sqrt(0.5)), True, 16),
    (3, 16, 1, 1, "lrelu", True, 256, 1.0, True, 8),                              # fromrgb on an fp16 block
    (3, 16, 1, 1, "lrelu", True, 256, 1.0, False, 8),
]


@pytest.mark.parametrize("case", CONV)
def test_conv2d_layer_fused_equals_composed(case, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    cin, cout, k, down, act, bias, clamp, gain, half, res = case
    layer = N.Conv2dLayer(cin, cout, kernel_size=k, bias=bias, activation=act, down=down, conv_clamp=clamp)
    _init(layer, 5)
    x = _rnd((2, cin, res, res), 5).contiguous(memory_format=torch.channels_last)
    if half:
        x = x.half()
    _check(lambda x: layer(x, gain=gain), layer, [x], half, monkeypatch)


@pytest.mark.parametrize("act,bias,lrm,n,fin,fout", [("linear", True, 1.0, 4, 24, 16), ("lrelu", True, 0.01, 3, 40, 24), ("linear", False, 1.0, 2, 8, 1),
                                                   ("lrelu", True, 1.0, 16, 512, 300)])
def test_fully_connected_fused_equals_composed(act, bias, lrm, n, fin, fout, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.FullyConnectedLayer(fin, fout, bias=bias, activation=act, lr_multiplier=lrm, bias_init=0.5)
    _init(layer, 6)
    _check(lambda x: layer(x), layer, [_rnd((n, fin), 5)], False, monkeypatch)


def test_second_order_phases_keep_the_composed_operators(emu):
    """outside first_order() with autograd on, the layers differentiate twice: the fused (once-differentiable) nodes must not be used"""
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.SynthesisLayer(16, 16, w_dim=24, resolution=8, conv_clamp=256)
    _init(layer, 3)
    x = _rnd((2, 16, 8, 8), 5).requires_grad_(True)
    w = _rnd((2, 24), 6).requires_grad_(True)
    y = layer(x, w, noise_mode="const", fused_modconv=False)
    (gw,) = torch.autograd.grad(y.square().sum(), w, create_graph=True)
    gw.square().sum().backward()
    assert x.grad is not None and layer.weight.grad is not None


def test_prepared_weights_follow_the_parameter(emu):
    """the prepared weight forms are rebuilt when the parameter changes in place (optimiser step) and by refresh() in one call"""
    from ic_gan_amd import ops
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    layer = N.Conv2dLayer(16, 16, kernel_size=3, activation="lrelu")
    x = _rnd((1, 16, 8, 8), 5)
    calls = []
    orig = ops.sg2_weight_prep_multi
    ops.sg2_weight_prep_multi = lambda items: (calls.append(len(items)), orig(items))[1]
    try:
        with torch.no_grad():
            y0 = layer(x)
            y1 = layer(x)
            assert calls == [1] and torch.equal(y0, y1)
            layer.weight.mul_(2.0)
            FL.refresh(layer)
            assert calls == [1, 1]
            y2 = layer(x)
            assert calls == [1, 1]
            ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, layer.weight * layer.weight_gain, padding=1) + layer.bias.view(1, -1, 1, 1),
                                                 0.2) * np.sqrt(2)
            assert float((y2 - ref).abs().max()) < 1e-4
    finally:
        ops.sg2_weight_prep_multi = orig


# ---------------------------------------------------------------------------------------------- second order (path-length regulariser)
def _second_order(fn, params, inputs, fused):
    """the path-length pattern of loss.py:112-146 on one layer: g = d <y, r> / d w with create_graph, then the gradient of |g|^2 + <g, q>"""
    from ic_gan_amd.stylegan_ops import conv2d_gradfix, fused_layers as FL
    import contextlib
    for p in params:
        p.grad = None
    ins = [t.detach().clone().requires_grad_(True) for t in inputs]
    with (FL.second_order() if fused else contextlib.nullcontext()):
        y = fn(*ins)
        # the incoming gradient dy = r * rs carries history (in a network it is the next layer's dx): its cotangent is checked through rs
        rs = _rnd(tuple(y.shape), 95).requires_grad_(True)
        ins.append(rs)
        r = _rnd(tuple(y.shape), 99)
        with conv2d_gradfix.no_weight_gradients():
            # (both the style gradient -- the path-length vector -- and the gradient that travels on to the previous layer)
            g, gx = torch.autograd.grad([(y.float() * (r * rs)).sum()], [ins[1], ins[0]], create_graph=True, only_inputs=True)
        q, qx, qy = _rnd(tuple(g.shape), 98), _rnd(tuple(gx.shape), 97), _rnd(tuple(y.shape), 96)
        # ... and a cotangent on the layer's OUTPUT in the final pass, as the next layer's second-order node sends one
        (g.square().sum() * 0.5 + (g * q).sum() + (gx.float() * qx).sum() + (y.float() * qy).sum() * 0.1).backward()
    return g.detach().float(), [t.grad.float() for t in ins], [p.grad.float() if p.grad is not None else None for p in params]


def _check2(fn, mod, inputs, half, monkeypatch):
    from ic_gan_amd import _lib as L
    params, names = list(mod.parameters()), [n for n, _ in mod.named_parameters()]
    seen = []
    orig = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (seen.append(name), orig(name, *a))[1])
    g0, gi0, gp0 = _second_order(fn, params, inputs, fused=False)
    assert not any(n.startswith("icg_sg2_") for n in seen)
    del seen[:]
    g1, gi1, gp1 = _second_order(fn, params, inputs, fused=True)
    assert any(n in ("icg_sg2_act_bwd2", "icg_sg2_torgb_bwd2") for n in seen), "second-order fused path not taken: %s" % sorted(set(seen))
    tol = 1.5e-2 if half else 1e-4
    _cmp(g1, g0, 4e-3 if half else 2e-5, "first-order gradient (the path-length vector)")
    for i, (a, b) in enumerate(zip(gi1, gi0)):
        _cmp(a, b, tol, "second-order grad input %d" % i)
    for n, a, b in zip(names, gp1, gp0):
        if n.endswith("noise_strength") or (n.endswith("bias") and "affine" not in n):
            continue          # (their second-order gradients are identically zero: masks only)
        _cmp(a, b, tol, "second-order grad " + n)


@pytest.mark.parametrize("case", SYN)
def test_synthesis_layer_second_order_fused_equals_composed(case, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    cin, cout, res, up, half, noise_mode, clamp, gain, n = case
    layer = N.SynthesisLayer(cin, cout, w_dim=24, resolution=res, up=up, conv_clamp=clamp)
    _init(layer, 3)
    draws = _rnd((n, 1, res, res), 77)
    monkeypatch.setattr(N, "_randn", lambda shape, device: draws.clone())
    x = _rnd((n, cin, res // up, res // up), 5).contiguous(memory_format=torch.channels_last)
    w = _rnd((n, 24), 6)
    if half:
        x = x.half()
    _check2(lambda x, w: layer(x, w, noise_mode=noise_mode, fused_modconv=False, gain=gain), layer, [x, w], half, monkeypatch)


@pytest.mark.parametrize("half,clamp,with_img,cin", [(False, None, False, 16), (True, 256, True, 32), (True, 0.7, True, 64), (False, 0.5, True, 512)])
def test_torgb_layer_second_order_fused_equals_composed(half, clamp, with_img, cin, emu, monkeypatch):
    from ic_gan_amd.stylegan2 import networks as N
    layer = N.ToRGBLayer(cin, 3, w_dim=24, conv_clamp=clamp)
    _init(layer, 4)
    n, res = 2, 8
    x = _rnd((n, cin, res, res), 5).contiguous(memory_format=torch.channels_last)
    w = _rnd((n, 24), 6)
    img = _rnd((n, 3, res, res), 8)
    if half:
        x = x.half()
    if with_img:
        _check2(lambda x, w, img: layer(x, w, fused_modconv=False, img=img), layer, [x, w, img], half, monkeypatch)
    else:
        _check2(lambda x, w: layer(x, w, fused_modconv=False), layer, [x, w], half, monkeypatch)


def test_second_order_chain_of_layers_with_relaid_inputs(emu):
    """two layers in a row fed the way the networks feed them -- x in NCHW order (re-laid-out inside the node), w an unbound slice of ws (made
    contiguous inside the node): the backward nodes must receive the layers' OWN inputs, or the cotangents of x and ws are dropped on the way"""
    import contextlib
    from ic_gan_amd.stylegan2 import networks as N
    from ic_gan_amd.stylegan_ops import conv2d_gradfix, fused_layers as FL
    l1 = N.SynthesisLayer(16, 16, w_dim=24, resolution=8, conv_clamp=256)
    l2 = N.SynthesisLayer(16, 16, w_dim=24, resolution=8, up=1, conv_clamp=256)
    _init(l1, 3); _init(l2, 4)
    x0, ws0, r = _rnd((2, 16, 8, 8), 1), _rnd((2, 2, 24), 2), _rnd((2, 16, 8, 8), 3)

    def run(fused):
        for p in list(l1.parameters()) + list(l2.parameters()):
            p.grad = None
        x, ws = x0.clone().requires_grad_(True), ws0.clone().requires_grad_(True)
        with (FL.second_order() if fused else contextlib.nullcontext()):
            w1, w2 = ws.unbind(1)
            y = l2(l1(x, w1, noise_mode="const", fused_modconv=False), w2, noise_mode="const", fused_modconv=False)
            with conv2d_gradfix.no_weight_gradients():
                (g,) = torch.autograd.grad([(y * r).sum()], [ws], create_graph=True, only_inputs=True)
            (g.square().sum(2).mean(1).sqrt() - 0.5).square().sum().backward()
        return [x.grad.clone(), ws.grad.clone(), l1.weight.grad.clone(), l2.affine.weight.grad.clone()]

    for what, a, b in zip(("x", "ws", "l1.weight", "l2.affine.weight"), run(True), run(False)):
        _cmp(a, b, 1e-4, what)

"""CPU-only: exercises the product's HOST logic (module wiring, autograd shims, layouts, step schedule)
with the C-ABI calls routed to oracle/kernel_ref.py (a test-only monkeypatch; the product has no such
route) and checks it end-to-end against the reference-generated goldens.  What this pins: every formula
the Python side composes out of kernel contracts (fused BN backward, SN backward, OHWI/HWIO/dgrad layouts,
commuted shortcuts, attention decomposition, Adam/EMA bookkeeping).  The kernels themselves are checked
on the GPU (tests/test_kernels_gpu.py, tests/test_parity_gpu.py)."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref, synth
from tests.helpers import CASES, GRAD_RTOL, STATE_RTOL, adam_slack, check_group, load_golden


@pytest.fixture
def emu(monkeypatch):
    kernel_ref.install(monkeypatch)


def _build(g):
    import ic_gan_amd.BigGAN as M
    cfg = g["cfg"]
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    return M, G, D


@pytest.mark.parametrize("case", CASES)
def test_state_dict_contract(case):
    """names, order and shapes of state_dict() equal the reference's (checkpoint drop-in)."""
    g = load_golden(case)
    _, G, D = _build(g)
    assert synth.spec_of(G.state_dict()) == g["gspec"]
    assert synth.spec_of(D.state_dict()) == g["dspec"]
    assert G.dim_z == int(g["dim_z"])


@pytest.mark.parametrize("case", ["cc_ic_r64", "cc_r32_flat", "cc_ic_r128"])
def test_forward_host_logic(case, emu):
    g = load_golden(case)
    cfg = g["cfg"]
    _, G, D = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    D.load_state_dict(synth.synth_state(g["dspec"], 22))
    G.train(); D.train()
    c = synth.CondSampler(cfg, G.dim_z, int(g["g_batch"]), seed=5)()
    z = c[0] if isinstance(c, tuple) else c
    lab = c[1] if cfg["class_cond"] else None
    fg = c[-1] if cfg["instance_cond"] else None
    with torch.no_grad():
        img = G(z, lab, fg)
        logit = D(img, lab, fg)
    if "fwd/img" in g:
        np.testing.assert_allclose(img.numpy(), g["fwd/img"], rtol=2e-4, atol=5e-5)
    np.testing.assert_allclose(logit.numpy(), g["fwd/logit"], rtol=2e-4, atol=2e-4)
    check_group(g, "fwd/G_state/", G.state_dict(), rtol=1e-4, atol=1e-6, what="G buf ")
    check_group(g, "fwd/D_state/", D.state_dict(), rtol=1e-4, atol=1e-6, what="D buf ")


def test_attention_stacked_projections_equal_the_three_convolutions(emu, monkeypatch):
    """ops.AttnProjFn (theta / phi / g as one 1x1 convolution with stacked weights + icg_attn_split_pool) against the layer-by-layer
    form (three SNConv2d + two max-pools, reference layers.py:217-231): same outputs, same gradients of x, of the four weights and
    gamma, same power-iteration state -- on the emulated kernels, where both forms are exact up to fp32 summation order."""
    import copy
    import functools
    from ic_gan_amd import layers, ops
    torch.manual_seed(3)
    conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, num_svs=1, num_itrs=1, eps=1e-6)
    a = layers.Attention(32, conv)
    with torch.no_grad():
        a.gamma.fill_(0.4)
    b = copy.deepcopy(a)
    x = torch.randn(2, 32, 8, 8)
    dy = torch.randn(2, 32, 8, 8)
    out = []
    for m, on in ((a, True), (b, False)):
        monkeypatch.setattr(ops, "FUSED_ATTENTION_PROJECTIONS", on)
        monkeypatch.setattr(ops, "FUSED_ATTENTION_OUTPUT", on)       # gamma folded into the output projection (ops.AttnOutFn)
        m.train()
        xi = x.clone().requires_grad_(True)
        h = xi * 1.5                       # a producer node, so that the chained gradient of x is summed inside the block
        y = m(h)
        y.backward(dy)
        out.append((y.detach(), xi.grad, {n: p.grad for n, p in m.named_parameters()}, {n: v.clone() for n, v in m.named_buffers()}))
    (ya, gxa, gpa, ba), (yb, gxb, gpb, bb) = out
    torch.testing.assert_close(ya, yb, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gxa, gxb, rtol=1e-4, atol=1e-6)
    assert set(gpa) == set(gpb) and all(g is not None for g in gpa.values())
    for n in gpa:
        torch.testing.assert_close(gpa[n], gpb[n], rtol=1e-4, atol=1e-6 + 1e-5 * float(gpb[n].abs().max()), msg=n)
    for n in ba:
        torch.testing.assert_close(ba[n], bb[n], rtol=1e-6, atol=1e-7, msg=n)
    # no-grad forward (G's pass of the D step) takes the stacked form too and needs no dgrad layouts
    with torch.no_grad():
        monkeypatch.setattr(ops, "FUSED_ATTENTION_PROJECTIONS", True)
        monkeypatch.setattr(ops, "FUSED_ATTENTION_OUTPUT", True)
        ya = a(x)
        monkeypatch.setattr(ops, "FUSED_ATTENTION_PROJECTIONS", False)
        monkeypatch.setattr(ops, "FUSED_ATTENTION_OUTPUT", False)
        torch.testing.assert_close(ya, b(x), rtol=1e-5, atol=1e-6)
    # gamma = 0 (the initial value): the block is the identity, its parameters still receive gradients
    for m, on in ((a, True), (b, False)):
        monkeypatch.setattr(ops, "FUSED_ATTENTION_OUTPUT", on)
        with torch.no_grad():
            m.gamma.zero_()
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = m(xi * 1.0)
        torch.testing.assert_close(y.detach(), x, rtol=0, atol=0)
        y.backward(dy)
        out.append((xi.grad, m.gamma.grad.clone(), m.o.weight.grad.clone()))
    (gx0, gg0, go0), (gx1, gg1, go1) = out[-2:]
    torch.testing.assert_close(gx0, gx1, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gg0, gg1, rtol=1e-4, atol=1e-6)
    assert float(gg1.abs()) > 0 and float(go0.abs().max()) == 0 and float(go1.abs().max()) == 0


@pytest.mark.parametrize("case", ["cc_ic_r64", "ic_r64_acc2"])
def test_prefetched_start_of_the_next_step_is_the_same_training_run(case, emu, monkeypatch):
    """train_fns.PREFETCH_NEXT_STEP issues the NEXT step's first host-side draw and generator forward at the end of the current step
    (behind the EMA, before the loss read-back): the same operations in the same order, so a run of steps is bit-identical with and
    without it -- losses of every step, every parameter of G and D and every buffer of D after the last one (G's buffers are one forward
    ahead by construction: running statistics and power-iteration vectors of the forward already issued) -- and the sampler is asked
    exactly one extra time (the draw waiting for a step that never came).  A stash whose generator was touched in between is dropped."""
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    g = load_golden(case)
    cfg = g["cfg"]
    gb, steps = int(g["g_batch"]), 3
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    runs = []
    for prefetch in (False, True):
        monkeypatch.setattr(train_fns, "PREFETCH_NEXT_STEP", prefetch)
        M, G, D = _build(g)
        G.load_state_dict(synth.synth_state(g["gspec"], 11))
        D.load_state_dict(synth.synth_state(g["dspec"], 22))
        G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
        ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
        opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
        opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
        GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
        state = {"itr": 0}
        inner = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
        calls = []
        samp = lambda: (calls.append(1), inner())[1]
        train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False, device="cpu", batch_size=gb)
        losses = []
        for s in range(steps):
            x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
            state["itr"] += 1
            G.train(); D.train(); G_ema.train()
            losses.append(train(x, y, f))
        runs.append((losses, [p.detach().clone() for p in list(G.parameters()) + list(D.parameters()) + list(D.buffers())], len(calls)))
    (l0, p0, c0), (l1, p1, c1) = runs
    assert l0 == l1
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))
    assert c1 == c0 + 1, (c0, c1)


@pytest.mark.parametrize("wino", [0, 2, 4, 5])
@pytest.mark.parametrize("case", ["cc_ic_r64", "ic_r64_acc2", "cc_r32_flat"])
def test_train_step_host_logic(case, emu, wino, monkeypatch):
    if wino:      # force the Winograd F(2x2,3x3) form onto every eligible 3x3 layer of these narrow test networks
        import ic_gan_amd.ops as _ops
        monkeypatch.setattr(_ops, "WINOGRAD_MIN_CHANNELS", 4)
        monkeypatch.setattr(_ops, "WINOGRAD2_MIN_CHANNELS", 4)
        monkeypatch.setattr(_ops, "WINOGRAD4_MIN_CHANNELS", 4 if wino >= 4 else 10 ** 9)      # F(4x4,3x3) / F(2x2,3x3)
        monkeypatch.setattr(_ops, "WINOGRAD4_WGRAD_MIN_CHANNELS", 4 if wino >= 4 else 10 ** 9)
        if wino == 5:     # ... and the resample-fused layers (GBlock conv1, DBlock conv2) in the 25-plane domain, all directions
            monkeypatch.setattr(_ops, "RS_WINOGRAD_MIN_CHANNELS", {True: (4, 4, 4), False: (4, 4, 4)})
    from ic_gan_amd import train_fns, utils
    from ic_gan_amd.optim import FusedAdam
    g = load_golden(case)
    cfg = g["cfg"]
    M, G, D = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    D.load_state_dict(synth.synth_state(g["dspec"], 22))
    G_ema = M.Generator(**{**cfg, "skip_init": True, "no_optim": True})
    ema = utils.ema(G, G_ema, cfg["ema_decay"], cfg["ema_start"])
    opt_d = FusedAdam(D.parameters(), lr=cfg["D_lr"], betas=(cfg["D_B1"], cfg["D_B2"]), eps=cfg["adam_eps"])
    opt_g = FusedAdam(G.parameters(), lr=cfg["G_lr"], betas=(cfg["G_B1"], cfg["G_B2"]), eps=cfg["adam_eps"])
    GD = M.G_D(G, D, optimizer_G=opt_g, optimizer_D=opt_d)
    state = {"itr": 0}
    gb, steps = int(g["g_batch"]), int(g["steps"])
    samp = synth.CondSampler(cfg, G.dim_z, gb, seed=7)
    train = train_fns.GAN_training_function(G, D, GD, ema, state, cfg, samp, embedded_optimizers=False,
                                            device="cpu", batch_size=gb)
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    for s in range(steps):
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
        state["itr"] += 1
        G.train(); D.train(); G_ema.train()
        m = train(x, y, f)
        np.testing.assert_allclose([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]], g["losses"][s],
                                   rtol=5e-4, atol=5e-4)
        if s == 0:
            check_group(g, "step1/G_grad/", {n: p.grad for n, p in G.named_parameters() if p.grad is not None},
                        GRAD_RTOL, 1e-6, "G grad ")
            check_group(g, "step1/D_grad/", {n: p.grad for n, p in D.named_parameters() if p.grad is not None},
                        GRAD_RTOL, 1e-6, "D grad ")
        gx = adam_slack(g, "step1/G_grad/", cfg["G_lr"], s + 1, G.state_dict().keys())
        dx = adam_slack(g, "step1/D_grad/", cfg["D_lr"], s + 1, D.state_dict().keys())
        check_group(g, f"step{s + 1}/G_state/", G.state_dict(), STATE_RTOL, 2e-6, "G ", extra_atol=gx)
        check_group(g, f"step{s + 1}/D_state/", D.state_dict(), STATE_RTOL, 2e-6, "D ", extra_atol=dx)
        check_group(g, f"step{s + 1}/EMA_state/", G_ema.state_dict(), STATE_RTOL, 2e-6, "EMA ", extra_atol=gx)


def test_grouped_spectral_norm_backward_equals_the_per_layer_one(emu, monkeypatch):
    """ops.SNGroupFn: from the second forward on (layouts known, batched prefetch) the layers' autograd nodes hand the RAW gradient of
    W / sigma to one node per group of layers, whose backward runs the spectral-norm backward for the group -- same parameter
    gradients as the per-layer path (SN_BACKWARD_GROUP = 0), through G and D (3x3 / 1x1 / upsample- and pool-fused convolutions,
    linears, the class embedding, the attention block's stacked projections), and torch.autograd.grad works on the weights."""
    import copy
    from ic_gan_amd import layers, ops
    g = load_golden("cc_ic_r64")
    cfg = g["cfg"]
    _, G, D = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    D.load_state_dict(synth.synth_state(g["dspec"], 22))
    z, lab, fg = synth.CondSampler(cfg, G.dim_z, 2, seed=5)()
    nets = {}
    groups = []
    real_many = ops.sn_backward_many
    monkeypatch.setattr(ops, "sn_backward_many", lambda items: (groups.append(len(items)), real_many(items))[1])
    for tag, size in (("grouped", 4), ("per_layer", 0)):
        monkeypatch.setattr(ops, "SN_BACKWARD_GROUP", size)
        Gn, Dn = copy.deepcopy(G), copy.deepcopy(D)
        Gn.train(); Dn.train()
        for it in range(2):                       # the first pass records the layouts; the second one is prefetched (and grouped)
            for p in list(Gn.parameters()) + list(Dn.parameters()):
                p.grad = None
            n0 = len(groups)
            out = Dn(Gn(z, lab, fg), lab, fg)
            out.sum().backward()
            if tag == "grouped":
                assert (len(groups) > n0) == (it == 1)
        nets[tag] = (Gn, Dn)
    assert not any(isinstance(m, layers.SN) and m._sn_ready is not None for n in nets["grouped"] for m in n.modules())
    n_sn = sum(1 for n in (G, D) for m in n.modules() if isinstance(m, layers.SN) and m.weight.requires_grad)
    assert sum(groups) == n_sn and max(groups) <= 4
    for a, b in zip(nets["grouped"], nets["per_layer"]):
        for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
            assert (p.grad is None) == (q.grad is None), k
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), k
        for (k, v), (_, v2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert torch.equal(v, v2), k
    # the functional form: gradients of a few weights only (a third pass, grouped)
    monkeypatch.setattr(ops, "SN_BACKWARD_GROUP", 4)
    Gn, Dn = nets["grouped"]
    ws = [Gn.blocks[0][0].conv1.weight, Dn.blocks[0][0].conv2.weight, Gn.linear.weight]
    ref = [w.grad.clone() for w in ws]
    got = torch.autograd.grad(Dn(Gn(z, lab, fg), lab, fg).sum(), ws)
    for a, b in zip(got, ref):
        assert a.shape == b.shape and torch.isfinite(a).all()


def test_grouped_conditional_bn_projections_equal_the_four_linears(emu, monkeypatch):
    """ops.CcbnAffineFn (bn1.gain / bn1.bias / bn2.gain / bn2.bias of a GBlock on the same y as one grouped launch per direction,
    reference layers.py:367-374) against the four SNLinear calls: same image, same gradients of every parameter and of the shared
    embedding, same spectral-norm buffers."""
    import copy
    from ic_gan_amd import ops
    g = load_golden("cc_ic_r64")
    cfg = g["cfg"]
    _, G, _ = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    z, lab, fg = synth.CondSampler(cfg, G.dim_z, 16, seed=5)()      # (the grouped form starts at 16 rows)
    calls = []
    real = ops._linear_group
    monkeypatch.setattr(ops, "_linear_group", lambda mode, M, K, items: (calls.append((mode, len(items))), real(mode, M, K, items))[1])
    res = []
    for on in (True, False):
        monkeypatch.setattr(ops, "GROUPED_CCBN", on)
        Gn = copy.deepcopy(G)
        Gn.train()
        for it in range(2):
            for p in Gn.parameters():
                p.grad = None
            img = Gn(z, lab, fg)
            (img * torch.linspace(-1, 1, img.numel()).view_as(img)).sum().backward()
        res.append((img.detach(), Gn))
    nblk = sum(1 for st in G.blocks for m in st if hasattr(m, "bn1"))
    assert calls.count((0, 4)) == 2 * nblk and calls.count((1, 4)) == 2 * nblk and calls.count((2, 4)) == 2 * nblk
    (ia, Ga), (ib, Gb) = res
    torch.testing.assert_close(ia, ib, rtol=1e-4, atol=3e-5)        # (two different torch matmul routes on the emulated kernels)
    # the two forms differ by fp32 summation order (here: two torch matmul routes), which this ReLU network amplifies to ~1e-4 of a
    # gradient tensor's norm; gradients that are mathematically zero (a conv bias in front of a BatchNorm) are rounding noise in both
    top = max(float(q.grad.norm()) for q in Gb.parameters() if q.grad is not None)
    for (k, p), (_, q) in zip(Ga.named_parameters(), Gb.named_parameters()):
        assert (p.grad is None) == (q.grad is None), k
        if p.grad is not None:
            assert float((p.grad - q.grad).norm()) <= 1e-3 * float(q.grad.norm()) + 1e-6 * top, k
    for (k, v), (_, v2) in zip(Ga.state_dict().items(), Gb.state_dict().items()):
        if "weight" not in k:
            torch.testing.assert_close(v, v2, rtol=1e-5, atol=1e-6, msg=k)


def test_training_passes_leave_no_reference_cycles(emu):
    """A forward + backward pass must free its graph by reference counting alone: a cycle through an autograd node (e.g. node ->
    SNState -> handle -> node) keeps every saved activation alive until Python's cycle collector runs, which does not see GPU memory
    (the cfg3 bench ran out of 288 GB after 25 steps that way).  With the collector switched off the number of live tensors must not
    grow from pass to pass."""
    import gc
    g = load_golden("cc_ic_r64")
    cfg = g["cfg"]
    _, G, D = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    D.load_state_dict(synth.synth_state(g["dspec"], 22))
    G.train(); D.train()
    z, lab, fg = synth.CondSampler(cfg, G.dim_z, 2, seed=5)()

    def one_pass():
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
        D(G(z, lab, fg), lab, fg).sum().backward()

    def live():
        return sum(1 for o in gc.get_objects() if type(o) is torch.Tensor or type(o) is torch.nn.Parameter)      # (no isinstance: it pokes lazy objects)

    one_pass(); one_pass()                       # layouts recorded, grouped paths active
    gc.collect()
    gc.disable()
    try:
        one_pass()
        n0 = live()
        for _ in range(3):
            one_pass()
        n1 = live()
    finally:
        gc.enable()
    assert n1 <= n0 + 4, (n0, n1)


def test_sn_prefetch_bookkeeping(emu):
    """the batched spectral-norm pass is used from the second forward on, gives the same buffers as the per-layer path,
    and a forward that aborts midway leaves the module usable (prefetch is transactional)."""
    import copy
    from ic_gan_amd import layers
    g = load_golden("cc_ic_r64")
    cfg = g["cfg"]
    _, G, D = _build(g)
    G.load_state_dict(synth.synth_state(g["gspec"], 11))
    G2 = copy.deepcopy(G)
    c = synth.CondSampler(cfg, G.dim_z, 2, seed=5)()
    z, lab, fg = c
    calls = []
    real = layers.ops.sn_prepare_many
    layers.ops.sn_prepare_many = lambda items, eps, training: (calls.append(len(items)), real(items, eps, training))[1]
    try:
        with torch.no_grad():
            G.train(); G2.train()
            a1 = G(z, lab, fg)
            assert calls == []                      # layouts not known yet: per-layer path
            a2 = G(z, lab, fg)
            assert calls and sum(calls) == sum(1 for m in G.modules() if isinstance(m, layers.SN))
    finally:
        layers.ops.sn_prepare_many = real
    noop = layers.sn_prefetch
    layers.sn_prefetch = lambda modules: None
    try:
        with torch.no_grad():
            b1 = G2(z, lab, fg)
            b2 = G2(z, lab, fg)
    finally:
        layers.sn_prefetch = noop
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    for (k, v), (_, v2) in zip(G.state_dict().items(), G2.state_dict().items()):
        assert torch.equal(v, v2), k
    # an aborted forward must not brick the module: unconsumed prefetched states are dropped by the network's
    # try/finally; a state left behind by a bare sn_prefetch call is dropped with a warning at the next prefetch
    boom = G.blocks[1][0].conv2.forward
    G.blocks[1][0].conv2.forward = lambda *a, **k: (_ for _ in ()).throw(ValueError("abort mid-forward"))
    with torch.no_grad():
        with pytest.raises(ValueError):
            G(z, lab, fg)
    G.blocks[1][0].conv2.forward = boom
    assert all(m._sn_ready is None for m in G.modules() if isinstance(m, layers.SN))
    with torch.no_grad():
        out = G(z, lab, fg)                       # usable again
        assert torch.isfinite(out).all()
        layers.sn_prefetch([G.linear])            # prefetched, never consumed ...
        with pytest.warns(RuntimeWarning, match="never consumed"):
            G(z, lab, fg)                         # ... dropped with a warning, forward completes


def test_eval_mode_sn_cache_follows_weight_updates(emu):
    _eval_cache_case("cpu")


@pytest.mark.gpu
def test_eval_mode_sn_cache_follows_weight_updates_hip():
    _eval_cache_case("cuda:0")


def _eval_cache_case(dev):
    """eval-mode W/sigma is cached; the cache must be invalidated by everything that writes weights or u0 through raw
    pointers (fused Adam, EMA, the training-mode power iteration) and by load_state_dict."""
    from ic_gan_amd import layers, utils
    from ic_gan_amd.optim import FusedAdam
    g = load_golden("cc_ic_r64")
    cfg = g["cfg"]
    _, G, D = _build(g)
    G = G.to(dev)
    G.load_state_dict({k: v.to(dev) for k, v in synth.synth_state(g["gspec"], 11).items()})
    z, lab, fg = (t.to(dev) for t in synth.CondSampler(cfg, G.dim_z, 2, seed=5)())
    calls = {"n": 0}
    real_one, real_many = layers.ops.sn_prepare, layers.ops.sn_prepare_many

    def one(*a, **k):
        calls["n"] += 1
        return real_one(*a, **k)

    layers.ops.sn_prepare = one
    layers.ops.sn_prepare_many = lambda items, eps, training: [one(w, u, sv, eps, training, nd, up, dn, *rest) for (w, u, sv, nd, up, dn, *rest) in items]
    try:
        G.eval()
        with torch.no_grad():
            a0 = G(z, lab, fg)
            n0 = calls["n"]
            G(z, lab, fg)
            assert calls["n"] > n0                                  # the cache is OPT-IN: off by default
            # a `.data` write (the reference's utils.ema.update) bumps no version counter: without the opt-in it must still
            # be seen by the next eval forward
            G.linear.weight.data.mul_(1.5)
            assert not torch.equal(G(z, lab, fg), a0)
            G.linear.weight.data.div_(1.5)
        layers.enable_sn_eval_cache(G)
        with torch.no_grad():
            a = G(z, lab, fg)
            n1 = calls["n"]
            b = G(z, lab, fg)
            assert calls["n"] == n1 and torch.equal(a, b)          # second eval forward: everything from the cache
            G.linear.weight.data.mul_(1.5)                         # opted in + `.data` write: explicit invalidation hook
            layers.invalidate_sn_cache(G)
            assert not torch.equal(G(z, lab, fg), a)
            G.linear.weight.data.div_(1.5)
            layers.invalidate_sn_cache(G)
            a = G(z, lab, fg)
            n1 = calls["n"]
        opt = FusedAdam(G.parameters(), lr=1e-2, betas=(0.0, 0.999), eps=1e-6)
        G.train()
        G(z, lab, fg).sum().backward()
        opt.step()
        G.eval()
        with torch.no_grad():
            c = G(z, lab, fg)
        assert calls["n"] > n1 and not torch.equal(a, c)            # Adam + power iteration invalidated the cache
        G2 = copy_of(G)
        with torch.no_grad():
            assert torch.equal(G2(z, lab, fg), c)                   # a fresh module (no cache) agrees
        # EMA writes through raw pointers as well
        G_ema = layers.enable_sn_eval_cache(copy_of(G))
        with torch.no_grad():
            before = G_ema(z, lab, fg)
        e = utils.ema(G, G_ema, 0.5, 0)
        with torch.no_grad():
            G.linear.weight.add_(0.05)
        e.update(1)
        with torch.no_grad():
            after = G_ema(z, lab, fg)
        assert not torch.equal(before, after)
    finally:
        layers.ops.sn_prepare, layers.ops.sn_prepare_many = real_one, real_many


def copy_of(m):
    import copy
    c = copy.deepcopy(m)
    for mod in c.modules():
        if hasattr(mod, "_sn_eval"):
            mod._sn_eval = None
    return c


def test_winograd_routing_rules(monkeypatch):
    """which form a 3x3 layer takes is a pure function of its channel counts and (full) resolution -- never of the batch
    size, because the spectral-norm layouts are prefetched from the previous call (D sees batch 2B and B alternately)."""
    import ic_gan_amd.ops as ops
    # plain 3x3 stride-1 layers
    assert ops.winograd_applies(96, 96, 256, 256, 64) == 4 and ops.winograd_applies(96, 96, 256, 256, 1) == 4
    assert ops.winograd_applies(64, 96, 256, 256, 64) == 0                   # below the channel threshold
    assert ops.winograd_applies(192, 192, 6, 6, 64) == 2                     # not a multiple of 4: F(2x2,3x3) from 192 channels
    assert ops.winograd_applies(96, 96, 6, 6, 64) == 0
    assert ops.winograd_applies(96, 98, 8, 8, 64) == 0                       # channel quads only
    assert ops.winograd_wgrad_tile(96, 192, 128, 128, 128) == 4
    # resample-fused layers (h, w: full resolution)
    for batch in (1, 64, 128):
        assert ops.resample_winograd_applies(192, 96, 256, 256, batch) == 5
    assert ops.resample_winograd_applies(192, 96, 6, 6, 64) == 0
    assert ops.resample_winograd_directions(192, 96, True) == (True, True, True)       # upsample-fused: fprop / dgrad / wgrad (round 3: dgrad too)
    assert ops.resample_winograd_directions(192, 64, True) == (False, False, False)
    assert ops.resample_winograd_directions(96, 96, False) == (True, True, False)      # pool-fused: forward / data gradient from 96 channels
    assert ops.resample_winograd_directions(64, 96, False) == (False, False, False)    # (round 4, fused narrow-layer kernel), weight gradient from 192
    assert ops.resample_winograd_directions(384, 192, False) == (True, True, True)
    # the strict route
    for k in ("WINOGRAD_MIN_CHANNELS", "WINOGRAD2_MIN_CHANNELS", "WINOGRAD4_MIN_CHANNELS", "WINOGRAD4_WGRAD_MIN_CHANNELS",
              "RS_WINOGRAD_MIN_CHANNELS"):
        monkeypatch.setattr(ops, k, getattr(ops, k))                          # restored after the test
    ops.disable_winograd()
    assert ops.winograd_applies(1536, 1536, 8, 8, 64) == 0 and ops.winograd_wgrad_tile(1536, 1536, 8, 8, 64) == 0
    assert ops.resample_winograd_applies(1536, 1536, 8, 8, 64) == 0


def test_bench_accounts_every_convolution_entry_point():
    """bench.py's KernelTimer must know every convolution entry point the host code can dispatch (a missing one would silently
    drop its FLOPs and time from the roofline table)."""
    import importlib.util
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    src = open(os.path.join(root, "ic_gan_amd", "ops.py")).read()
    called = set(re.findall(r'L\.call\("(icg_conv2d_[a-z0-9_]+)"', src))
    called |= {"icg_conv2d_%s_%s" % (v, k) for v in ("wino", "wino4") for k in ("fprop", "wgrad")}      # "%s" call sites
    called = {n for n in called if "%" not in n}
    missing = sorted(called - set(bench.KernelTimer.SPEC))
    assert not missing, missing

"""StyleGAN2 custom ops: parity with goldens produced by the reference's own `impl='ref'`
implementations (tests/golden/make_golden_stylegan_ops.py), forward + first- and second-order gradients.
CPU part pins the per-kernel oracle (oracle/kernel_ref.py); GPU part checks the HIP kernels through the product's
`ic_gan_amd.stylegan_ops` wrappers."""
import os

import numpy as np
import pytest
import torch

from oracle import kernel_ref as R
from tests.stylegan_cases import ACTS, UPFIR, rnd

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stylegan_ops.npz"))
ACT_ID = {a: i + 1 for i, a in enumerate(ACTS)}
DEF_GAIN = dict(linear=1, relu=np.sqrt(2), lrelu=np.sqrt(2), tanh=1, sigmoid=1, elu=1, selu=1, softplus=1, swish=np.sqrt(2))


@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("ci", [0, 1])
def test_kernel_ref_bias_act_pinned_to_reference(act, ci):
    ai = ACTS.index(act)
    clamp = -1.0 if ci == 0 else 0.7
    x, b, dy = rnd((3, 6, 5, 5), 10 + ai, 1.5), rnd((6,), 20 + ai, 0.5), rnd((3, 6, 5, 5), 30 + ai)
    n = x.numel()
    alpha, gain = (0.2 if act == "lrelu" else 0.0), float(DEF_GAIN[act])
    y = torch.empty(n)
    R.icg_bias_act(x, b, None, None, None, y, n, 25, 6, 0, ACT_ID[act], alpha, gain, clamp)
    np.testing.assert_allclose(y.view(3, 6, 5, 5).numpy(), G[f"ba/{act}/{ci}/y"], rtol=1e-5, atol=1e-6)
    dx = torch.empty(n)
    R.icg_bias_act(dy, b, x, y.view_as(x), None, dx, n, 25, 6, 1, ACT_ID[act], alpha, gain, clamp)
    np.testing.assert_allclose(dx.view(3, 6, 5, 5).numpy(), G[f"ba/{act}/{ci}/dx"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("i", range(len(UPFIR)))
def test_kernel_ref_upfirdn2d_pinned_to_reference(i):
    n, c, h, w, taps, up, down, pad, flip, gain = UPFIR[i]
    x = rnd((n, c, h, w), 50 + i)
    f = torch.from_numpy(G[f"up/{i}/f"])
    f2 = f.ger(f) if f.ndim == 1 else f
    yg = G[f"up/{i}/y"]
    y = torch.empty(yg.shape)
    R.icg_upfirdn2d(x, f2.contiguous(), y, n, c, h, w, f2.shape[0], f2.shape[1], up, up, down, down, pad[0], pad[1],
                    pad[2], pad[3], int(flip), gain, yg.shape[2], yg.shape[3])
    np.testing.assert_allclose(y.numpy(), yg, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("ci", [0, 1])
def test_bias_act_hip_matches_reference(act, ci):
    from ic_gan_amd.stylegan_ops import bias_act
    ai = ACTS.index(act)
    clamp = None if ci == 0 else 0.7
    x = rnd((3, 6, 5, 5), 10 + ai, 1.5).cuda().requires_grad_(True)
    b = rnd((6,), 20 + ai, 0.5).cuda().requires_grad_(True)
    dy = rnd((3, 6, 5, 5), 30 + ai).cuda().requires_grad_(True)
    d2 = rnd((3, 6, 5, 5), 40 + ai).cuda()
    y = bias_act.bias_act(x, b, act=act, clamp=clamp)
    k = f"ba/{act}/{ci}/"
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[k + "y"], rtol=1e-5, atol=1e-6)
    dx, db = torch.autograd.grad(y, (x, b), dy, create_graph=True)
    np.testing.assert_allclose(dx.detach().cpu().numpy(), G[k + "dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(db.detach().cpu().numpy(), G[k + "db"], rtol=1e-4, atol=1e-4)
    ddx, ddy = torch.autograd.grad((dx * d2).sum(), (x, dy), allow_unused=True)
    ddx = torch.zeros_like(x) if ddx is None else ddx
    np.testing.assert_allclose(ddy.detach().cpu().numpy(), G[k + "ddy"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ddx.detach().cpu().numpy(), G[k + "ddx"], rtol=2e-3, atol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(len(UPFIR)))
def test_upfirdn2d_hip_matches_reference(i):
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    n, c, h, w, taps, up, down, pad, flip, gain = UPFIR[i]
    x = rnd((n, c, h, w), 50 + i).cuda().requires_grad_(True)
    f = U.setup_filter(taps, device="cuda")
    np.testing.assert_allclose(f.cpu().numpy(), G[f"up/{i}/f"], rtol=1e-6, atol=1e-7)
    y = U.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[f"up/{i}/y"], rtol=1e-4, atol=1e-5)
    dy = rnd(tuple(y.shape), 60 + i).cuda()
    (dx,) = torch.autograd.grad(y, x, dy)
    np.testing.assert_allclose(dx.cpu().numpy(), G[f"up/{i}/dx"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_upfirdn2d_wrappers_hip():
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    x = rnd((2, 3, 8, 8), 70).cuda()
    f = U.setup_filter([1, 3, 3, 1], device="cuda")
    for name in ("upsample2d", "downsample2d", "filter2d"):
        np.testing.assert_allclose(getattr(U, name)(x, f).cpu().numpy(), G["wrap/" + name], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("i", [i for i, c in enumerate(UPFIR) if c[1] % 4 == 0])
def test_upfirdn2d_channels_last_hip(i):
    """icg_upfirdn2d_nhwc (channels-last in and out) against the same reference goldens."""
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    n, c, h, w, taps, up, down, pad, flip, gain = UPFIR[i]
    x = rnd((n, c, h, w), 50 + i).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    f = U.setup_filter(taps, device="cuda")
    y = U.upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip, gain=gain)
    assert y.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(y.detach().cpu().numpy(), G[f"up/{i}/y"], rtol=1e-4, atol=1e-5)
    dy = rnd(tuple(y.shape), 60 + i).cuda().contiguous(memory_format=torch.channels_last)
    (dx,) = torch.autograd.grad(y, x, dy)
    np.testing.assert_allclose(dx.cpu().numpy(), G[f"up/{i}/dx"], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,up,down,pad", [((2, 8, 9, 7), 1, 1, [2, 1, 1, 2]), ((1, 16, 8, 8), 2, 1, [2, 1, 2, 1]),
                                                ((2, 4, 13, 13), 1, 2, [1, 1, 1, 1]), ((1, 12, 5, 6), 2, 2, [3, 0, 0, 3]),
                                                ((3, 8, 33, 17), 1, 1, [0, 0, 0, 0])])
def test_upfirdn2d_nhwc_equals_nchw_kernel(shape, up, down, pad):
    """strip / boundary handling of the channels-last kernel at shapes the goldens do not cover."""
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    x = rnd(shape, 7).cuda()
    f = U.setup_filter([1, 3, 3, 1], device="cuda")
    a = U.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=1.7)
    b = U.upfirdn2d(x.contiguous(memory_format=torch.channels_last), f, up=up, down=down, padding=pad, gain=1.7)
    assert not a.is_contiguous(memory_format=torch.channels_last) or a.shape[1] == 1
    torch.testing.assert_close(b.contiguous(), a, rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float16, torch.float64])
@pytest.mark.parametrize("shape,taps,pad,flip", [((2, 64, 33, 33), [1, 3, 3, 1], [1, 1, 1, 1], False),      # blur after an up-sampling conv
                                                 ((1, 128, 16, 24), [1, 3, 3, 1], [2, 2, 2, 2], True),       # ... before a down-sampling one (adjoint: flipped)
                                                 ((2, 64, 9, 7), [1, 2, 1], [1, 1, 1, 1], False),            # 3 x 3 filter, a single ragged tile
                                                 ((1, 192, 40, 17), [1, 3, 3, 1], [3, 0, 0, 3], False),      # three channel groups, one-sided padding
                                                 ((3, 64, 5, 5), [1, 3, 3, 1], [0, 0, 0, 0], False)])        # 'valid': output 2 x 2
def test_upfirdn2d_channels_last_lds_tile(dtype, shape, taps, pad, flip):
    """upfirdn2d_nhwc_tile_kernel (csrc/stylegan_ops_typed.hip: channels-last FIR with up = down = 1 through an LDS tile, fp16 / fp64
    storage) against the NCHW kernel on the same data: same taps in the same order, so equal up to the storage rounding."""
    from ic_gan_amd.stylegan_ops import upfirdn2d as U
    x = rnd(shape, 11).cuda().to(dtype)
    f = U.setup_filter(taps, device="cuda")
    a = U.upfirdn2d(x, f, padding=pad, flip_filter=flip, gain=1.3)
    b = U.upfirdn2d(x.contiguous(memory_format=torch.channels_last), f, padding=pad, flip_filter=flip, gain=1.3)
    assert b.dtype == dtype and b.is_contiguous(memory_format=torch.channels_last) and a.shape == b.shape
    # (fp64: the channels-last kernels fold the gain into their fp32 filter taps, the NCHW kernel multiplies the fp64 sum by it)
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(b.contiguous().double(), a.double(), **tol)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(x.double(), pad), (f if flip else f.flip([0, 1])).double()[None, None].repeat(
        shape[1], 1, 1, 1) * 1.3, groups=shape[1])
    torch.testing.assert_close(b.contiguous().double(), ref, rtol=3e-3 if dtype == torch.float16 else 1e-6,
                               atol=3e-3 if dtype == torch.float16 else 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("act", ACTS)
@pytest.mark.parametrize("cl", [False, True])
def test_bias_act_vectorised_paths(act, cl):
    """the 16-byte kernels (plane-major bias for NCHW, channel-minor for channels-last) against oracle/kernel_ref.py,
    which test_kernel_ref_bias_act_pinned_to_reference pins to the reference; forward, first and second order."""
    from ic_gan_amd.stylegan_ops import bias_act as B
    from oracle import kernel_ref as K
    shape = (2, 8, 6, 6)
    x = rnd(shape, 1, 1.5)
    b = rnd((8,), 2, 0.5)
    dy, d2 = rnd(shape, 3), rnd(shape, 4)
    xg = x.cuda()
    if cl:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg = xg.requires_grad_(True)
    bg = b.cuda().requires_grad_(True)
    dyg = dy.cuda().requires_grad_(True)
    y = B.bias_act(xg, bg, act=act, clamp=0.9)
    assert y.is_contiguous(memory_format=torch.channels_last) == cl or not cl
    dx, db = torch.autograd.grad(y, (xg, bg), dyg, create_graph=True)
    ddx, ddy = torch.autograd.grad((dx * d2.cuda()).sum(), (xg, dyg), allow_unused=True)
    # CPU reference through the (scalar, pinned) emulation of the same entry point
    act_id, alpha, gain = B.activation_funcs[act][0], B.activation_funcs[act][1], B.activation_funcs[act][2]
    n, step = x.numel(), 36
    yr, g1, g2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    K.icg_bias_act(x, b, None, None, None, yr, n, step, 8, 0, act_id, alpha, gain, 0.9)
    K.icg_bias_act(dy, b, x, yr, None, g1, n, step, 8, 1, act_id, alpha, gain, 0.9)
    torch.testing.assert_close(y.detach().cpu().contiguous(), yr, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dx.detach().cpu().contiguous(), g1, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(db.detach().cpu(), g1.sum([0, 2, 3]), rtol=1e-4, atol=1e-4)
    K.icg_bias_act(d2, b, x, yr, None, g2, n, step, 8, 1, act_id, alpha, gain, 0.9)          # d(dx.d2)/d(dy)
    torch.testing.assert_close(ddy.detach().cpu().contiguous(), g2, rtol=1e-4, atol=1e-5)
    if ddx is not None:
        K.icg_bias_act(d2, b, x, yr, dy, g2, n, step, 8, 2, act_id, alpha, gain, 0.9)
        torch.testing.assert_close(ddx.detach().cpu().contiguous(), g2, rtol=1e-3, atol=1e-4)


def _nan_to_num_tensors(dev):
    """gradient-like tensors: vector-path sizes (multiples of 4096), ragged tails, a scalar, an empty one, > 64 tensors (two launches)"""
    sizes = [4096, 8192 + 3, 1, 0, 5000, 12288, 17] + [33 + k for k in range(70)]
    out = []
    for i, n in enumerate(sizes):
        t = rnd((n,), 300 + i, 3.0)
        if n:
            k = max(n // 7, 1)
            t[::k] = float("nan")
            t[1::k + 1] = float("inf")
            t[2::k + 2] = float("-inf")
        out.append(t.to(dev))
    return out


@pytest.mark.parametrize("dev", ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)])
@pytest.mark.parametrize("args", [dict(nan=0, posinf=1e5, neginf=-1e5), dict(nan=2.5)])
def test_nan_to_num_multi_equals_torch(dev, args, monkeypatch):
    """ops.nan_to_num_multi (icg_nan_to_num_multi: every gradient of a network in one launch per 64 tensors) against
    torch.nan_to_num per tensor -- the reference's loop, training_loop.py:511-515 -- bit for bit, NaN / +inf / -inf in the vector and
    the ragged parts of the tensors; None = torch's defaults (the largest finite fp32)."""
    import ic_gan_amd.ops as ops
    if dev == "cpu":
        R.install(monkeypatch)
    ts = _nan_to_num_tensors(dev)
    want = [torch.nan_to_num(t.cpu(), **args) for t in ts]
    versions = [t._version for t in ts]
    ops.nan_to_num_multi(ts, **args)
    for t, w, v in zip(ts, want, versions):
        assert torch.equal(t.cpu(), w)
        assert t.numel() == 0 or t._version > v

"""The StyleGAN2 plugin seam: `ic_gan_amd.stylegan_ops.plugin.get_plugin(...)` must be usable
exactly like the module `torch_utils.custom_ops.get_plugin` returns in the reference.

The GPU tests drive it with the reference's OWN call pattern -- what `_bias_act_cuda` (ops/bias_act.py:231-317) and
`_upfirdn2d_cuda` (ops/upfirdn2d.py:268-349) do with `_plugin.bias_act` / `_plugin.upfirdn2d`: empty tensors for absent
arguments, grad = 0 / 1 / 2 passes fed by saved x / y / dy, the backward of upfirdn2d as another upfirdn2d -- and compare with
goldens produced by the unmodified reference's `impl='ref'` implementations (tests/golden/stylegan_ops.npz for fp32,
stylegan_ops_typed.npz for fp16 / fp64 storage; make_golden_stylegan_ops.py).
Tolerances: fp32 2e-5 of max|ref|; fp64 2e-7 (the ABI's scalars are C floats); fp16 one storage rounding of the result plus one of each saved tensor the
pass reads (y is re-read in its fp16-rounded form by the gradient passes, as in the CUDA plugin): 4e-3 of max|ref|."""
import os

import numpy as np
import pytest
import torch

from tests.stylegan_cases import ACTS, UPFIR, rnd

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ACT_SPEC = {  # name: (cuda_idx, def_alpha, def_gain, ref, has_2nd_grad)   -- bias_act.py:25-106
    "linear": (1, 0.0, 1.0, "", False), "relu": (2, 0.0, np.sqrt(2), "y", False), "lrelu": (3, 0.2, np.sqrt(2), "y", False),
    "tanh": (4, 0.0, 1.0, "y", True), "sigmoid": (5, 0.0, 1.0, "y", True), "elu": (6, 0.0, 1.0, "y", True),
    "selu": (7, 0.0, 1.0, "y", True), "softplus": (8, 0.0, 1.0, "y", True), "swish": (9, 0.0, np.sqrt(2), "x", True),
}
# fp64: alpha / gain / clamp cross the plugin ABI as C `float` (bias_act.cpp:35, upfirdn2d.cpp:19) and the filter is fp32, while
# the `ref` goldens use Python doubles: sqrt(2) as float is 1.7e-8 off -> 2e-7
DT = {"f32": (torch.float32, 2e-5), "f16": (torch.float16, 4e-3), "f64": (torch.float64, 2e-7)}


def _close(got, ref, tol, what):
    got, ref = got.detach().double().cpu().numpy(), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(np.abs(ref).max()) + 1e-30
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all() and err <= tol * scale, f"{what}: max abs err {err:.3e} vs {tol:.0e} * {scale:.3e}"


def test_plugin_rejects_what_the_reference_plugin_rejects():
    """argument validation of bias_act.cpp:38-58 / upfirdn2d.cpp:22-31 -> RuntimeError (runs without a GPU: the first check of
    both plugins is `x must reside on CUDA device`)."""
    from ic_gan_amd.stylegan_ops import plugin
    x, e = torch.zeros(2, 3, 4, 4), torch.empty(0)
    with pytest.raises(RuntimeError, match="must reside on CUDA device"):
        plugin.bias_act(x, e, e, e, e, 0, 1, 1, 0.0, 1.0, -1.0)
    with pytest.raises(RuntimeError, match="must reside on CUDA device"):
        plugin.upfirdn2d(x, torch.ones(1, 1), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match="no plugin named"):
        plugin.get_plugin("filtered_lrelu_plugin")


@pytest.mark.gpu
def test_plugin_argument_checks_on_device():
    from ic_gan_amd.stylegan_ops import plugin
    p = plugin.get_plugin("bias_act_plugin", sources=["bias_act.cpp", "bias_act.cu"], extra_cuda_cflags=["--use_fast_math"])
    x, e = torch.zeros(2, 3, 4, 4, device="cuda"), torch.empty(0, device="cuda")
    for args, msg in [((x, torch.zeros(3, device="cuda", dtype=torch.float16), e, e, e, 0, 1, 1, 0., 1., -1.), "same dtype and device"),
                      ((x, torch.zeros(4, device="cuda"), e, e, e, 0, 1, 1, 0., 1., -1.), "wrong number of elements"),
                      ((x, torch.zeros(3, device="cuda"), e, e, e, 0, 7, 1, 0., 1., -1.), "dim is out of bounds"),
                      ((x, e, torch.zeros(2, 3, 4, 5, device="cuda"), e, e, 1, 1, 2, 0., 1., -1.), "xref must have the same shape"),
                      ((x, e, e, e, e, -1, 1, 1, 0., 1., -1.), "grad must be non-negative"),
                      ((x[:, :, ::2], e, e, e, e, 0, 1, 1, 0., 1., -1.), "non-overlapping and dense"),
                      ((x, e, e, x.contiguous(memory_format=torch.channels_last), e, 1, 1, 2, 0., 1., -1.), "yref must have the same layout")]:
        with pytest.raises(RuntimeError, match=msg):
            p.bias_act(*args)
    u = plugin.get_plugin("upfirdn2d_plugin")
    with pytest.raises(RuntimeError, match="f must be float32"):
        u.upfirdn2d(x, torch.ones(2, 2, device="cuda", dtype=torch.float64), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)
    with pytest.raises(RuntimeError, match="output must be at least 1x1"):
        u.upfirdn2d(x, torch.ones(4, 4, device="cuda"), 1, 1, 1, 1, -3, -3, 0, 0, False, 1.0)
    assert p.bias_act(torch.zeros(0, 3, device="cuda"), e, e, e, e, 0, 1, 1, 0., 1., -1.).shape == (0, 3)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dt", ["f32", "f16", "f64"])
@pytest.mark.parametrize("act", ACTS)
def test_bias_act_plugin_reference_call_pattern(act, dt, layout):
    from ic_gan_amd.stylegan_ops import plugin
    p = plugin.get_plugin("bias_act_plugin")
    dtype, tol = DT[dt]
    g = np.load(os.path.join(GOLD, "stylegan_ops.npz" if dt == "f32" else "stylegan_ops_typed.npz"))
    pre = "" if dt == "f32" else dt + "/"
    idx, alpha, gain, ref, has2 = ACT_SPEC[act]
    ai = ACTS.index(act)
    mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    dev = lambda t: t.to("cuda", dtype).contiguous(memory_format=mf) if t.dim() == 4 else t.to("cuda", dtype)
    null = torch.empty([0], device="cuda", dtype=dtype)
    for ci, clamp in enumerate([-1.0, 0.7]):
        x, b = dev(rnd((3, 6, 5, 5), 10 + ai, 1.5)), dev(rnd((6,), 20 + ai, 0.5))
        dy, d2 = dev(rnd((3, 6, 5, 5), 30 + ai)), dev(rnd((3, 6, 5, 5), 40 + ai))
        k = f"{pre}ba/{act}/{ci}/"
        # BiasActCuda.forward (bias_act.py:246-257)
        y = p.bias_act(x, b, null, null, null, 0, 1, idx, alpha, gain, clamp)
        assert y.dtype == dtype and y.stride() == x.stride()
        _close(y, g[k + "y"], tol, f"y {act} {dt} {layout} clamp {clamp}")
        if act == "linear" and ci == 1:
            continue          # the CUDA plugin does not mask the gradient of a clamped linear op (bias_act.py:262-266); `ref` does
        xs, bs = (x, b) if ("x" in ref or has2) else (null, null)
        ys = y if "y" in ref else null
        # BiasActCudaGrad.forward (bias_act.py:290-299)
        dx = p.bias_act(dy, bs, xs, ys, null, 1, 1, idx, alpha, gain, clamp)
        _close(dx, g[k + "dx"], tol, f"dx {act} {dt} {layout} clamp {clamp}")
        # BiasActCudaGrad.backward (bias_act.py:301-317): d_dy is the grad=1 pass on d_dx, d_x the grad=2 pass
        ddy = p.bias_act(d2, bs, xs, ys, null, 1, 1, idx, alpha, gain, clamp)
        _close(ddy, g[k + "ddy"], tol, f"ddy {act} {dt} {layout}")
        if has2:
            ddx = p.bias_act(d2, bs, xs, ys, dy, 2, 1, idx, alpha, gain, clamp)
            _close(ddx, g[k + "ddx"], 4 * tol, f"ddx {act} {dt} {layout}")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dt", ["f32", "f16", "f64"])
@pytest.mark.parametrize("ci", range(len(UPFIR)))
def test_upfirdn2d_plugin_reference_call_pattern(ci, dt, layout):
    from ic_gan_amd.stylegan_ops import plugin
    from ic_gan_amd.stylegan_ops.upfirdn2d import setup_filter
    p = plugin.get_plugin("upfirdn2d_plugin")
    dtype, tol = DT[dt]
    g = np.load(os.path.join(GOLD, "stylegan_ops.npz" if dt == "f32" else "stylegan_ops_typed.npz"))
    pre = "" if dt == "f32" else dt + "/"
    n, c, h, w, taps, up, down, pad, flip, gain = UPFIR[ci]
    mf = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    x = rnd((n, c, h, w), 50 + ci).to("cuda", dtype).contiguous(memory_format=mf)
    f = setup_filter(taps, flip_filter=False).cuda()
    f2 = (f.ger(f) if f.ndim == 1 else f).contiguous()        # the plugin takes rank-2 filters (upfirdn2d.cpp:29)
    px0, px1, py0, py1 = pad
    # Upfirdn2dCuda.forward (upfirdn2d.py:287-319; a separable filter = the same 2-D pass)
    y = p.upfirdn2d(x, f2, up, up, down, down, px0, px1, py0, py1, flip, gain)
    assert y.dtype == dtype
    assert y.is_contiguous(memory_format=mf)
    _close(y, g[f"{pre}up/{ci}/y"], tol, f"upfirdn2d y case {ci} {dt} {layout}")
    # Upfirdn2dCuda.backward (upfirdn2d.py:329-346): up <-> down, flipped filter, adjoint padding
    dy = rnd(tuple(y.shape), 60 + ci).to("cuda", dtype).contiguous(memory_format=mf)
    fh, fw = f2.shape
    oh, ow = y.shape[2:]
    q = [fw - px0 - 1, w * up - ow * down + px0 - up + 1, fh - py0 - 1, h * up - oh * down + py0 - up + 1]
    dx = p.upfirdn2d(dy, f2, down, down, up, up, q[0], q[1], q[2], q[3], not flip, gain)
    _close(dx, g[f"{pre}up/{ci}/dx"], tol, f"upfirdn2d dx case {ci} {dt} {layout}")

"""bias_act on the HIP kernel `icg_bias_act` — API of stylegan2_ada_pytorch/torch_utils/ops/bias_act.py:131-171.

    y = clamp(gain * act(x + b[dim]))        act in {linear, relu, lrelu, tanh, sigmoid, elu, selu, softplus, swish}

Autograd mirrors the reference plugin's scheme (bias_act.py:231-317): the first-order gradient is the same kernel
with grad=1 fed by the saved input/output, the second-order gradient (R1 / path-length regularisers,
training/loss.py:126-131,182-187) is grad=2.  fp16 / fp32 / fp64 storage with the plugin's internal arithmetic type (fp32,
fp64 for double: bias_act.cu:18-21); contiguous or channels-last tensors; no fallback implementation."""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib as L
from .. import ops as _ops

# name -> (plugin id, default alpha, default gain, what the backward needs: 'x', 'y' or '', has 2nd-order term)
activation_funcs = {
    "linear": (1, 0.0, 1.0, "", False),
    "relu": (2, 0.0, float(np.sqrt(2)), "y", False),
    "lrelu": (3, 0.2, float(np.sqrt(2)), "y", False),
    "tanh": (4, 0.0, 1.0, "y", True),
    "sigmoid": (5, 0.0, 1.0, "y", True),
    "elu": (6, 0.0, 1.0, "y", True),
    "selu": (7, 0.0, 1.0, "y", True),
    "softplus": (8, 0.0, 1.0, "y", True),
    "swish": (9, 0.0, float(np.sqrt(2)), "x", True),
}


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}

# Gradient of a CLAMPED act='linear' bias_act (ToRGB with conv_clamp).  False (default): the true gradient -- zero where the
# output was clamped -- which is what the reference's own `impl='ref'` path computes.  True: bit-compatible with the
# reference's CUDA plugin path, which does not keep y for act='linear' (bias_act.py:262-266) and therefore passes the
# gradient through unmasked.  Parity runs against a GPU-trained reference set this to True.
REFERENCE_CLAMP_GRAD = False


def _is_cl(t, dim):
    """4-D tensor stored channels-last (NHWC in memory) with the bias on the channel dim: handled in place, the kernel's
    (step_b, size_b) addressing covers it with step_b = 1 — no NCHW round trip between the NHWC convolutions."""
    return (t.ndim == 4 and dim == 1 and not t.is_contiguous()
            and t.is_contiguous(memory_format=torch.channels_last))


def _lay(t, cl):
    if t is None:
        return None
    return t.contiguous(memory_format=torch.channels_last) if cl else t.contiguous()


def _kernel(x, b, xref, yref, dy, grad, dim, act_id, alpha, gain, clamp):
    _ops._require_gpu(x)          # fails loudly off-GPU: there is no CPU path
    lead = next(t for t in (xref, yref, x) if t is not None)     # saved tensors fix the layout of a gradient pass
    cl = _is_cl(lead, dim)
    x, xref, yref, dy = _lay(x, cl), _lay(xref, cl), _lay(yref, cl), _lay(dy, cl)
    y = torch.empty_like(x)
    n = x.numel()
    if n == 0:
        return y
    step_b = (1 if cl else int(np.prod(x.shape[dim + 1:]))) if b is not None else 1
    size_b = b.numel() if b is not None else 1
    if x.dtype == torch.float32:
        L.call("icg_bias_act", x, b, xref, yref, dy, y, n, step_b, size_b, grad, act_id, float(alpha), float(gain),
               float(clamp))
    else:
        L.call("icg_bias_act_typed", x, b, xref, yref, dy, y, n, step_b, size_b, grad, act_id, float(alpha), float(gain),
               float(clamp), _DTYPES[x.dtype])
    return y


def bias_act(x, b=None, dim=1, act="linear", alpha=None, gain=None, clamp=None, impl="hip"):
    assert isinstance(x, torch.Tensor) and x.dtype in _DTYPES, "float16 / float32 / float64 storage"
    act_id, def_alpha, def_gain, ref, has2 = activation_funcs[act]
    alpha = float(alpha if alpha is not None else def_alpha)
    gain = float(gain if gain is not None else def_gain)
    clamp = float(clamp if clamp is not None else -1)
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        b = b.contiguous().to(x.dtype)       # the plugin requires b.dtype == x.dtype (bias_act.cpp:46); callers cast likewise
    return _BiasAct.apply(x, b, dim, act_id, alpha, gain, clamp, ref, has2)


class _BiasGrad(torch.autograd.Function):
    """db = dx.sum(all dims but `dim`) of bias_act.py:127 / 149 for fp16 channels-last tensors on icg_colsum_f16 (fp32 sums in a fixed
    order, rounded to the bias dtype once; torch's strided fp16 reduction took 30 - 80 us per layer and pass).  Linear in dx, so its
    backward is the broadcast -- differentiable again."""

    @staticmethod
    def forward(ctx, dx):
        ctx.shape = tuple(dx.shape)
        n, c, h, w = dx.shape
        rows = n * h * w
        nb = L.query("icg_colsum_f16_workspace_bytes", rows, c)
        out = torch.empty(c, device=dx.device, dtype=torch.float32)
        L.call("icg_colsum_f16", dx, rows, c, out, _ops._bytes(nb, dx.device), nb)
        return out.to(dx.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.reshape(1, -1, 1, 1).expand(ctx.shape)


def _bias_grad(dx, dim):
    if (FUSED_BIAS_GRAD and dx.dtype == torch.float16 and dx.is_cuda and _is_cl(dx, dim)
            and L.query("icg_colsum_f16_applies", int(dx.shape[1]))):
        return _BiasGrad.apply(dx)
    return dx.sum([i for i in range(dx.ndim) if i != dim])


FUSED_BIAS_GRAD = os.environ.get("ICG_FUSED_BIAS_GRAD", "1") != "0"        # measurement switch


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act_id, alpha, gain, clamp, ref, has2):
        x = _lay(x, _is_cl(x, dim))
        y = _kernel(x, b, None, None, None, 0, dim, act_id, alpha, gain, clamp)
        ctx.cfg = (dim, act_id, alpha, gain, clamp, ref, has2)
        # y is also kept whenever a clamp is active: the clamp mask of the gradient needs it.  (The reference's CUDA
        # plugin drops y for act='linear' (bias_act.py:262-266) and therefore does not mask the gradient of a clamped
        # linear bias_act — e.g. ToRGB with conv_clamp; we follow its own `impl='ref'` semantics, the true gradient.)
        keep_y = ("y" in ref) or (clamp >= 0 and not (REFERENCE_CLAMP_GRAD and act_id == 1))
        ctx.save_for_backward(x if ("x" in ref or has2) else None, b if ("x" in ref or has2) else None,
                              y if keep_y else None)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dim, act_id, alpha, gain, clamp, ref, has2 = ctx.cfg
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx = dy
            if act_id != 1 or gain != 1 or clamp >= 0:
                dx = _BiasActGrad.apply(dy, x, b, y, ctx.cfg)
        if ctx.has_b and ctx.needs_input_grad[1]:
            db = _bias_grad(dx, dim)
        return dx, db, None, None, None, None, None, None, None


class _BiasActGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, cfg):
        dim, act_id, alpha, gain, clamp, ref, has2 = cfg
        dx = _kernel(dy, b, x, y, None, 1, dim, act_id, alpha, gain, clamp)
        ctx.cfg = cfg
        ctx.save_for_backward(dy if has2 else None, x, b, y)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        dim, act_id, alpha, gain, clamp, ref, has2 = ctx.cfg
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _BiasActGrad.apply(d_dx, x, b, y, ctx.cfg)
        if has2 and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _kernel(d_dx, b, x, y, dy, 2, dim, act_id, alpha, gain, clamp)
        if has2 and ctx.needs_input_grad[2] and d_x is not None:
            d_b = _bias_grad(d_x, dim)
        return d_dy, d_x, d_b, None, None

"""The StyleGAN2 plugin seam: drop-in modules for what `torch_utils.custom_ops.get_plugin` returns.

The reference builds two JIT extensions and calls them through this interface
(stylegan2_ada_pytorch/torch_utils/custom_ops.py:52-148; callers ops/bias_act.py:165-171,231-317 and
ops/upfirdn2d.py:187-193,268-349):

    _plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> Tensor      (bias_act.cpp:35,97-100)
    _plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> Tensor
                                                                                              (upfirdn2d.cpp:19,101-104)

Same conventions here: a tensor with `numel() == 0` means "absent"; the output is allocated by the callee with the layout of
`x` (`empty_like` / `suggest_memory_format`); bad arguments raise RuntimeError with the plugin's messages (TORCH_CHECK);
the launch goes to the current stream of `x`'s device; fp16 / fp32 / fp64 storage with fp32 (fp64 for double) arithmetic
(bias_act.cu:18-21).  `get_plugin(module_name, ...)` mirrors the reference's entry point, so the one-line change in the
reference is `from ic_gan_amd.stylegan_ops.plugin import get_plugin` in torch_utils/custom_ops.py.
There is no reference-path fallback here: without the HIP library the call raises.
"""
from __future__ import annotations

import types

import torch

from .. import _lib as L

_INT_MAX = 2 ** 31 - 1
_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _same_layout(a, b):
    """bias_act.cpp:19-31"""
    if a.dim() != b.dim():
        return False
    for i in range(a.dim()):
        if a.size(i) != b.size(i):
            return False
        if a.size(i) >= 2 and a.stride(i) != b.stride(i):
            return False
    return True


def _is_dense_perm(t):
    """non-overlapping and dense = some permutation of a contiguous tensor"""
    dims = sorted(range(t.dim()), key=lambda i: (t.stride(i), t.size(i)))
    expect = 1
    for i in dims:
        if t.size(i) == 1:
            continue
        if t.stride(i) != expect:
            return False
        expect *= t.size(i)
    return True


def _ptr(t):
    return t if t.numel() else None


def _on_device(t) -> bool:
    """`x must reside on CUDA device` (bias_act.cpp:38, upfirdn2d.cpp:22).  A seam of its own so that the CPU binding test
    (tests/test_reference_binding_cpu.py: the reference's own callers over this module, kernels emulated) can lift it."""
    return t.is_cuda


def _device_of(t):
    return torch.cuda.device(t.device)


def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    """y = clamp(gain * act(x + b[dim])) (grad = 0), its first (1) or second (2) derivative pass — bias_act.cu:26-150."""
    _check(_on_device(x), "x must reside on CUDA device")
    _check(b.numel() == 0 or (b.dtype == x.dtype and b.device == x.device), "b must have the same dtype and device as x")
    for name, t in (("xref", xref), ("yref", yref)):
        _check(t.numel() == 0 or (t.shape == x.shape and t.dtype == x.dtype and t.device == x.device),
               f"{name} must have the same shape, dtype, and device as x")
    _check(dy.numel() == 0 or (dy.shape == x.shape and dy.dtype == x.dtype and dy.device == x.device),
           "dy must have the same dtype and device as x")
    _check(x.numel() <= _INT_MAX, "x is too large")
    _check(b.dim() == 1, "b must have rank 1")
    _check(b.numel() == 0 or (0 <= dim < x.dim()), "dim is out of bounds")
    _check(b.numel() == 0 or b.numel() == x.size(dim), "b has wrong number of elements")
    _check(grad >= 0, "grad must be non-negative")
    _check(_is_dense_perm(x), "x must be non-overlapping and dense")
    _check(b.is_contiguous(), "b must be contiguous")
    for name, t in (("xref", xref), ("yref", yref), ("dy", dy)):
        _check(t.numel() == 0 or _same_layout(t, x), f"{name} must have the same layout as x")
    _check(x.dtype in _DTYPES, "no kernel found for the tensor dtype (float16, float32, float64)")
    _check(1 <= int(act) <= 9 and grad <= 2, "no CUDA kernel found for the specified activation func")
    y = torch.empty_like(x)                  # preserve_format: same strides as the dense x
    _check(_same_layout(y, x), "y must have the same layout as x")
    if x.numel() == 0:
        return y
    step_b = int(x.stride(dim)) if b.numel() else 1
    with _device_of(x):
        L.call("icg_bias_act_typed", x, _ptr(b), _ptr(xref), _ptr(yref), _ptr(dy), y, x.numel(), step_b,
               max(int(b.numel()), 1), int(grad), int(act), float(alpha), float(gain), float(clamp), _DTYPES[x.dtype])
    return y


def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """zero-insert upsample -> pad / crop -> 2-D FIR -> decimate -> gain, per channel — upfirdn2d.cu:32-203."""
    _check(_on_device(x), "x must reside on CUDA device")
    _check(f.device == x.device, "f must reside on the same device as x")
    _check(f.dtype == torch.float32, "f must be float32")
    _check(x.numel() <= _INT_MAX, "x is too large")
    _check(f.numel() <= _INT_MAX, "f is too large")
    _check(x.dim() == 4, "x must be rank 4")
    _check(f.dim() == 2, "f must be rank 2")
    _check(f.size(0) >= 1 and f.size(1) >= 1, "f must be at least 1x1")
    _check(upx >= 1 and upy >= 1, "upsampling factor must be at least 1")
    _check(downx >= 1 and downy >= 1, "downsampling factor must be at least 1")
    _check(x.dtype in _DTYPES, "no kernel found for the tensor dtype (float16, float32, float64)")
    n, c, h, w = x.shape
    fh, fw = int(f.size(0)), int(f.size(1))
    out_w = (w * upx + padx0 + padx1 - fw + downx) // downx
    out_h = (h * upy + pady0 + pady1 - fh + downy) // downy
    _check(out_w >= 1 and out_h >= 1, "output must be at least 1x1")
    vec = {torch.float32: 4, torch.float16: 8, torch.float64: 2}[x.dtype]
    want_cl = cl = (not x.is_contiguous()) and x.is_contiguous(memory_format=torch.channels_last)
    if cl and (c % vec or fh * fw > 256):
        x, cl = x.contiguous(), False          # the channels-last kernels take whole 16-byte channel groups
    elif not cl and not x.is_contiguous():
        x = x.contiguous()
    y = torch.empty((n, c, out_h, out_w), device=x.device, dtype=x.dtype,
                    memory_format=torch.channels_last if cl else torch.contiguous_format)
    _check(y.numel() <= _INT_MAX, "output is too large")
    with _device_of(x):
        L.call("icg_upfirdn2d_typed", x, f.contiguous(), y, n, c, h, w, fh, fw, int(upx), int(upy), int(downx), int(downy),
               int(padx0), int(padx1), int(pady0), int(pady1), int(bool(flip)), float(gain), out_h, out_w, _DTYPES[x.dtype],
               int(cl))
    if want_cl and not cl:
        y = y.contiguous(memory_format=torch.channels_last)      # the plugin returns x.suggest_memory_format() (upfirdn2d.cpp:38)
    return y


_PLUGINS = {
    "bias_act_plugin": types.SimpleNamespace(bias_act=bias_act, __name__="bias_act_plugin"),
    "upfirdn2d_plugin": types.SimpleNamespace(upfirdn2d=upfirdn2d, __name__="upfirdn2d_plugin"),
}


def get_plugin(module_name, sources=None, **build_kwargs):
    """custom_ops.get_plugin (custom_ops.py:52-148) without the JIT build: the kernels ship prebuilt in libicgan_hip.so.
    `sources` / `build_kwargs` are accepted and ignored.  Unknown plugin names raise, a missing library raises."""
    if module_name not in _PLUGINS:
        raise RuntimeError(f"ic_gan_amd provides no plugin named {module_name!r} (have {sorted(_PLUGINS)})")
    L.lib()                     # fail loudly here, like the reference's build step would
    return _PLUGINS[module_name]

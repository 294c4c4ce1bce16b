"""upfirdn2d on the HIP kernel `icg_upfirdn2d` — API of stylegan2_ada_pytorch/torch_utils/ops/upfirdn2d.py
(setup_filter 88-139, upfirdn2d 145-193, filter2d 359, upsample2d 392, downsample2d 441).

Zero-insert upsample -> pad/crop -> 2-D FIR -> decimate -> gain, per channel.  The backward is another
upfirdn2d with up/down swapped, the filter flipped and the adjoint padding (reference 329-346), so arbitrary-order
gradients work.  fp16 / fp32 / fp64 storage (fp32 / fp64 accumulation), NCHW or channels-last; separable 1-D filters are
expanded to their outer product."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib as L
from .. import ops as _ops


_DTYPES = {torch.float32: 0, torch.float16: 1, torch.float64: 2}


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return int(sx), int(sy)


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return int(px0), int(px1), int(py0), int(py1)


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device("cpu"), normalize=True, flip_filter=False, gain=1, separable=None):
    """Same contract as the reference's setup_filter (upfirdn2d.py:88-139)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _filter2d(f, device):
    if f is None:
        return torch.ones([1, 1], dtype=torch.float32, device=device)
    f = f.to(device=device, dtype=torch.float32)
    return (f.ger(f) if f.ndim == 1 else f).contiguous()


def _run(x, f2, up, down, padding, flip, gain):
    _ops._require_gpu(x)          # fails loudly off-GPU: there is no CPU path
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = padding
    n, c, h, w = x.shape
    fh, fw = f2.shape
    out_w = (w * upx + px0 + px1 - fw + downx) // downx
    out_h = (h * upy + py0 + py1 - fh + downy) // downy
    assert out_w >= 1 and out_h >= 1
    if x.dtype != torch.float32:
        # fp16 / fp64 storage (upfirdn2d.cu:208-344 templates the kernels on the dtype; accumulation in fp32 / fp64)
        vec = 8 if x.dtype == torch.float16 else 2
        cl = c % vec == 0 and fh * fw <= 256 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
        y = torch.empty((n, c, out_h, out_w), device=x.device, dtype=x.dtype,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        L.call("icg_upfirdn2d_typed", x if cl else x.contiguous(), f2, y, n, c, h, w, fh, fw, upx, upy, downx, downy, px0, px1,
               py0, py1, int(bool(flip)), float(gain), out_h, out_w, _DTYPES[x.dtype], int(cl))
        return y
    if c % 4 == 0 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last):
        # channels-last in, channels-last out: no layout change between the NHWC convolutions and the FIR resampling
        y = torch.empty((n, c, out_h, out_w), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        L.call("icg_upfirdn2d_nhwc", x, f2, y, n, c, h, w, fh, fw, upx, upy, downx, downy, px0, px1, py0, py1,
               int(bool(flip)), float(gain), out_h, out_w)
        return y
    y = torch.empty(n, c, out_h, out_w, device=x.device, dtype=torch.float32)
    L.call("icg_upfirdn2d", x.contiguous(), f2, y, n, c, h, w, fh, fw, upx, upy, downx, downy, px0, px1, py0, py1,
           int(bool(flip)), float(gain), out_h, out_w)
    return y


class _Upfirdn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f2, up, down, padding, flip, gain):
        ctx.cfg = (up, down, padding, flip, gain, x.shape)
        ctx.save_for_backward(f2)
        return _run(x, f2, up, down, padding, flip, gain)

    @staticmethod
    def backward(ctx, dy):
        (f2,) = ctx.saved_tensors
        (upx, upy), (downx, downy), (px0, px1, py0, py1), flip, gain, xs = ctx.cfg
        _, _, ih, iw = xs
        _, _, oh, ow = dy.shape
        fh, fw = f2.shape
        p = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1, fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _Upfirdn2d.apply(dy, f2, (downx, downy), (upx, upy), tuple(p), not flip, gain)
        return dx, None, None, None, None, None, None


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl="hip"):
    assert isinstance(x, torch.Tensor) and x.ndim == 4 and x.dtype in _DTYPES
    if f is not None:
        assert f.dtype == torch.float32 and not f.requires_grad
    up, down, padding = _parse_scaling(up), _parse_scaling(down), _parse_padding(padding)
    # a 1-D (separable) filter f stands for f (x) f with gain split over the two passes: identical to one 2-D pass
    return _Upfirdn2d.apply(x, _filter2d(f, x.device), up, down, padding, bool(flip_filter), float(gain))


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl="hip"):
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + fw // 2, padx1 + (fw - 1) // 2, pady0 + fh // 2, pady1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl="hip"):
    upx, upy = _parse_scaling(up)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw + upx - 1) // 2, padx1 + (fw - upx) // 2, pady0 + (fh + upy - 1) // 2, pady1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl="hip"):
    downx, downy = _parse_scaling(down)
    padx0, padx1, pady0, pady1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [padx0 + (fw - downx + 1) // 2, padx1 + (fw - downx) // 2, pady0 + (fh - downy + 1) // 2,
         pady1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain)

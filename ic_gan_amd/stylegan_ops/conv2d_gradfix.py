"""conv2d / conv_transpose2d with arbitrary-order gradients on the HIP implicit-GEMM kernel.

API of stylegan2_ada_pytorch/torch_utils/ops/conv2d_gradfix.py (conv2d 43-64, conv_transpose2d 67-99,
no_weight_gradients 31-37); the reference builds it from cudnn_convolution / cudnn_convolution_transpose /
cudnn_convolution_backward_weight (139-272) so that R1 and path-length regularisation can differentiate twice.

Here every one of those contractions is one of two gathers of the C-ABI (include/icgan_hip.h):

    gather conv   G[geo](x, w)  = icg_conv2d_g_fprop   out[o] = sum_r src(o*stride + r - pad) w[r]
    gather wgrad  W[geo](x, dy) = icg_conv2d_g_wgrad   dw[r]  = sum_o x[o*stride + r - pad] dy[o]

and the family is closed under differentiation:
    d/dx  G[s, p, zins=0] = G[1, R-1-p, zins=s] with the flipped, transposed weight  (and back)
    d/dw  G               = W  (roles of x and dy swapped for the zero-inserted direction)
    d/dx  W(x, dy)        = adjoint gather of dy with the incoming cotangent as the weight;  d/ddy W = G(x, cotangent)
so both autograd Functions below express their backward with each other and gradients of any order exist.

Tensors are logical NCHW fp32 stored channels-last (NHWC in memory), weights are handed to the kernel as
[Cout][R][R][Cin].  dilation > 1 and non-square stride/padding are not supported (StyleGAN2 uses neither); groups > 1 — only
`fused_modconv`, off in training mode (networks.py:440-444) — runs as one groups=1 call per group (`_per_group`).

fp16 (the reference's `num_fp16_res` blocks, training/networks.py:77-91, 581-601: activations and weights cast to fp16, cuDNN
convolves in fp16 with fp32 accumulation): G runs on icg_conv2d_g_fprop_f16 -- fp16 operands straight into
v_mfma_f32_16x16x32_f16, fp32 accumulation, one rounding to fp16 -- whenever the kernel serves the shape (Cin % 32 == 0,
Cout % 64 == 0 or % 96 == 0; every 3x3 / 1x1 / stride-2 / transposed layer of those blocks and all their data gradients); the
3-channel toRGB / fromRGB layers and W (weight gradients) keep the exact-fp32 kernels between two casts.
"""
import contextlib
from dataclasses import dataclass

import torch

from .. import _lib as L
from .. import ops as _ops

enabled = True                       # kept for API parity; the HIP path is the only path
FP16_MFMA = True                     # fp16 blocks on the fp16-input MFMA kernel (False: exact-fp32 kernel between two casts; measurement switch)
weight_gradients_disabled = False    # conv2d_gradfix.py:26-37


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


@dataclass(frozen=True)
class _Geo:
    R: int
    stride: int      # gather stride over the source (1 when zins > 0)
    pad: int
    zins: int        # 0, or the zero-insertion factor of the source
    src: tuple       # (H, W) of the gathered tensor
    out: tuple       # (H, W) of the result

    def adjoint(self):
        """geometry of the data gradient (which gathers from `out`-shaped dy and produces `src`-shaped dx)."""
        p = self.R - 1 - self.pad
        if p < 0:
            raise NotImplementedError("padding larger than kernel_size - 1 is not supported")
        if self.zins:
            return _Geo(self.R, self.zins, p, 0, self.out, self.src)
        return _Geo(self.R, 1, p, self.stride if self.stride > 1 else 0, self.out, self.src)


_REVERSED_TAPS = {}    # (device, R*R) -> [R*R-1, ..., 0]


def _adjoint_weight(w):
    """[Cout][R][R][Cin] -> the weight of the adjoint gather: [Cin][R][R][Cout], taps flipped.  Flipping both tap axes reverses the
    flattened tap index, so transpose + flip is ONE gather along that axis of the transposed view (index_select writes a contiguous
    result): one launch per data gradient instead of flip + copy (156 -> 78 launches per cfg4 iteration, a host-bound loop)."""
    Cout, R, _, Cin = (int(v) for v in w.shape)
    key = (w.device, R * R)
    idx = _REVERSED_TAPS.get(key)
    if idx is None:
        idx = _REVERSED_TAPS[key] = torch.arange(R * R - 1, -1, -1, device=w.device)
    return w.reshape(Cout, R * R, Cin).permute(2, 1, 0).index_select(1, idx).view(Cin, R, R, Cout)


_PHASE_TAPS = {}      # device -> index tensor (made once: a host list -> device copy is not capturable in a HIP graph)


def _phase_weights(w):
    """[Cout][3][3][Cin] (gather form) -> wp[al][be][Cout][2][2][Cin] of icg_conv2d_tr2_fprop: an even output coordinate
    meets taps {0, 2}, an odd one tap {1} (second slot zero)."""
    w4 = torch.nn.functional.pad(w, (0, 0, 0, 1, 0, 1))                  # tap index 3 = zero
    idx = _PHASE_TAPS.get(w.device)                                      # (al, u) -> tap: (0,0)->0 (0,1)->2 (1,0)->1 (1,1)->zero
    if idx is None:
        idx = _PHASE_TAPS[w.device] = torch.tensor([0, 2, 1, 3], device=w.device)
    g = w4.index_select(1, idx).index_select(2, idx)                     # [Cout][al,u][be,v][Cin]
    Cout, Cin = w.shape[0], w.shape[3]
    return g.view(Cout, 2, 2, 2, 2, Cin).permute(1, 3, 0, 2, 4, 5).contiguous()


def gather_conv(x, w, geo, cache=None):
    """G[geo](x, w) as plain kernel calls (no autograd): x logical NCHW (fp16 / fp32), w [Cout][R][R][Cin].  `cache`: a dict owned by
    the caller that lives as long as `w` keeps its values -- derived weight forms (phase weights, Winograd transforms) are kept
    there instead of being rebuilt per call (stylegan_ops/fused_layers.py prepares a layer's weights once per optimiser step)."""
    _ops._require_gpu(x)
    half = x.dtype == torch.float16
    if half and w.dtype == torch.float16 and FP16_MFMA and \
            L.query("icg_conv2d_g_fprop_f16_applies", int(x.shape[1]), int(w.shape[0]), geo.R, geo.stride, geo.zins):
        # the reference's fp16 arithmetic: fp16 operands, fp32 accumulation, one rounding (csrc/hconv.hip)
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous()
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        assert (H, W) == geo.src and w.shape == (Cout, geo.R, geo.R, Cin), (x.shape, w.shape, geo)
        y = torch.empty((B, Cout, geo.out[0], geo.out[1]), device=x.device, dtype=torch.float16,
                        memory_format=torch.channels_last)
        L.call("icg_conv2d_g_fprop_f16", x, w, y, B, H, W, Cin, geo.out[0], geo.out[1], Cout, geo.R, geo.stride, geo.pad,
               geo.zins)
        return y
    if half:      # shapes the fp16 kernel does not serve: exact-fp32 kernel between two casts (layout glue)
        return _forward_f32(_ops._cl(x), _f32_weight(w, cache), geo, cache).to(torch.float16)
    # the fp32 kernels read raw fp32 buffers: both operands are normalised here (an fp16 weight next to fp32 activations would
    # otherwise be reinterpreted); fp64 operands are computed in fp32
    return _forward_f32(_ops._cl(x), _f32_weight(w, cache), geo, cache)


def _f32_weight(w, cache):
    if w.dtype == torch.float32 and w.is_contiguous():
        return w
    if cache is None:
        return w.float().contiguous()
    key = ("f32", w.data_ptr())
    if key not in cache:
        cache[key] = w.float().contiguous()
    return cache[key]


def _forward_f32(x, w, geo, cache=None):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    assert (H, W) == geo.src and w.shape == (Cout, geo.R, geo.R, Cin), (x.shape, w.shape, geo)
    y = _ops._empty_cl(B, Cout, geo.out[0], geo.out[1], x.device)
    if geo.zins == 2 and geo.R == 3 and geo.pad == 2 and geo.out[0] <= 2 * H + 2 and geo.out[1] <= 2 * W + 2:
        # stride-2 transposed 3x3 convolution: 4 phases of 2x2 taps instead of a gather over the zero-inserted source
        key = ("phase", w.data_ptr())
        wp = cache.get(key) if cache is not None else None
        if wp is None:
            wp = _phase_weights(w)
            if cache is not None:
                cache[key] = wp
        L.call("icg_conv2d_tr2_fprop", x, wp, None, y, B, H, W, Cin, geo.out[0], geo.out[1], Cout)
        return y
    if geo.R == 3 and geo.stride == 1 and geo.pad == 1 and geo.zins == 0 and geo.out == (H, W) and \
            _ops.winograd_applies(Cin, Cout, H, W, B):
        # wide 3x3 'same' convolution (synthesis conv1 / discriminator conv0 and the data gradients of both):
        # Winograd F(2x2,3x3), 16/36 of the multiply-adds and 16x the parallelism at low resolutions
        m = _ops.winograd_applies(Cin, Cout, H, W, B)
        v = "wino4" if m == 4 else "wino"
        key = (v, w.data_ptr())
        U = cache.get(key) if cache is not None else None
        if U is None:
            U = torch.empty((36 if m == 4 else 16) * Cout * Cin, device=x.device, dtype=torch.float32)
            L.call("icg_%s_weight_transform" % v, w, U, Cout, Cin)
            if cache is not None:
                cache[key] = U
        nbw = L.query("icg_conv2d_%s_workspace_bytes" % v, B, H, W, Cin, Cout)
        L.call("icg_conv2d_%s_fprop" % v, x, U, None, None, y, None, None, 0, B, H, W, Cin, Cout, 0, 1.0,
               _ops._bytes(nbw, x.device), nbw)
        return y
    nb = L.query("icg_conv2d_g_fprop_workspace_bytes", B, geo.out[0], geo.out[1], Cin, Cout, geo.R, geo.zins)
    if nb:    # too few output tiles to fill the chip: split-K
        L.call("icg_conv2d_g_fprop_ws", x, w, None, y, B, H, W, Cin, geo.out[0], geo.out[1], Cout, geo.R, geo.stride,
               geo.pad, geo.zins, _ops._bytes(nb, x.device), nb)
    else:
        L.call("icg_conv2d_g_fprop", x, w, None, y, B, H, W, Cin, geo.out[0], geo.out[1], Cout, geo.R, geo.stride,
               geo.pad, geo.zins)
    return y


class _GatherConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, geo):
        # the autograd inputs themselves are saved (not their re-laid-out copies): the backward differentiates through them
        ctx.geo = geo
        ctx.save_for_backward(x, w)
        return gather_conv(x, w, geo)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        geo = ctx.geo
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _GatherConv.apply(dy, _adjoint_weight(w), geo.adjoint())
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _GatherWgrad.apply(x, dy, geo)
        return dx, dw, None


def gather_wgrad_raw(x, dy, geo):
    """Weight gradient of y = G[geo](x, w) given dy, as the kernels write it (fp32, no autograd): (t, layout) with layout 0:
    t [R][R][Cin][Cout] = d w[co][r][s][ci];  layout 1 (zero-inserted direction -- dy is the gathered tensor): t [R][R][Cout][Cin] =
    d w[co][R-1-r][R-1-s][ci], i.e. indexed by the taps of the transposed convolution's own (un-flipped) weight."""
    _ops._require_gpu(x)
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    assert (H, W) == geo.src and tuple(dy.shape[2:]) == geo.out, (x.shape, dy.shape, geo)
    R = geo.R
    if x.dtype == torch.float16 and dy.dtype == torch.float16 and FP16_MFMA and \
            L.query("icg_conv2d_g_wgrad_f16_applies", int(Cin), int(Cout), R, geo.zins if geo.zins else geo.stride):
        # fp16 operands straight into the MFMA (csrc/hwgrad.hip); fp32 accumulation, fp32 HWIO result
        x = x.contiguous(memory_format=torch.channels_last)
        dy = dy.contiguous(memory_format=torch.channels_last)
        if geo.zins:      # zero-inserted direction: dy is the gathered tensor, x lives on the pixel grid
            adj = geo.adjoint()
            nb = L.query("icg_conv2d_g_wgrad_f16_workspace_bytes", B, H, W, Cout, Cin, R)
            t = torch.empty(R, R, Cout, Cin, device=x.device, dtype=torch.float32)
            L.call("icg_conv2d_g_wgrad_f16", dy, x, t, B, geo.out[0], geo.out[1], Cout, H, W, Cin, R, adj.stride, adj.pad,
                   _ops._bytes(nb, x.device), nb)
            return t, 1
        nb = L.query("icg_conv2d_g_wgrad_f16_workspace_bytes", B, geo.out[0], geo.out[1], Cin, Cout, R)
        t = torch.empty(R, R, Cin, Cout, device=x.device, dtype=torch.float32)
        L.call("icg_conv2d_g_wgrad_f16", x, dy, t, B, H, W, Cin, geo.out[0], geo.out[1], Cout, R, geo.stride, geo.pad,
               _ops._bytes(nb, x.device), nb)
        return t, 0
    x, dy = _ops._cl(x), _ops._cl(dy)
    if geo.zins:      # zero-inserted direction: dy is the gathered tensor, x lives on the pixel grid
        adj = geo.adjoint()
        ws_bytes = L.query("icg_conv2d_g_wgrad_workspace_bytes", B, H, W, Cout, Cin, R)
        t = torch.empty(R, R, Cout, Cin, device=x.device, dtype=torch.float32)
        L.call("icg_conv2d_g_wgrad", dy, x, t, B, geo.out[0], geo.out[1], Cout, H, W, Cin, R, adj.stride, adj.pad,
               _ops._bytes(ws_bytes, x.device), ws_bytes)
        return t, 1
    if R == 3 and geo.stride == 1 and geo.pad == 1 and geo.out == (H, W) and _ops.winograd_wgrad_tile(Cin, Cout, H, W, B):
        v = "wino4" if _ops.winograd_wgrad_tile(Cin, Cout, H, W, B) == 4 else "wino"      # Winograd-domain wgrad
        nbw = L.query("icg_conv2d_%s_wgrad_workspace_bytes" % v, B, H, W, Cin, Cout)
        t = torch.empty(R, R, Cin, Cout, device=x.device, dtype=torch.float32)
        L.call("icg_conv2d_%s_wgrad" % v, x, dy, t, None, None, 0, B, H, W, Cin, Cout, 0, _ops._bytes(nbw, x.device), nbw)
        return t, 0
    ws_bytes = L.query("icg_conv2d_g_wgrad_workspace_bytes", B, geo.out[0], geo.out[1], Cin, Cout, R)
    t = torch.empty(R, R, Cin, Cout, device=x.device, dtype=torch.float32)
    L.call("icg_conv2d_g_wgrad", x, dy, t, B, H, W, Cin, geo.out[0], geo.out[1], Cout, R, geo.stride, geo.pad,
           _ops._bytes(ws_bytes, x.device), ws_bytes)
    return t, 0


class _GatherWgrad(torch.autograd.Function):
    """dw[Cout][R][R][Cin] of y = G[geo](x, w) given dy."""

    @staticmethod
    def forward(ctx, x, dy, geo):
        ctx.geo = geo
        ctx.save_for_backward(x, dy)
        out_dtype = x.dtype                       # fp16 blocks: the gradient of an fp16 weight is fp16 (rounded once, from fp32)
        t, layout = gather_wgrad_raw(x, dy, geo)
        dw = t.flip(0, 1).permute(2, 0, 1, 3).contiguous() if layout else t.permute(3, 0, 1, 2).contiguous()
        return dw if out_dtype == torch.float32 else dw.to(out_dtype)

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        geo = ctx.geo
        gx = gdy = None
        if ctx.needs_input_grad[0]:
            gx = _GatherConv.apply(dy, _adjoint_weight(ddw), geo.adjoint())
        if ctx.needs_input_grad[1]:
            gdy = _GatherConv.apply(x, ddw.contiguous(), geo)
        return gx, gdy, None


class _Matmul(torch.autograd.Function):
    """C = op(A) op(B) on the HIP GEMM (icg_gemm_batched) for the dense layers of the mapping / affine / epilogue heads and the
    demodulation coefficients -- the reference calls cuBLAS here (torch.addmm / matmul, training/networks.py:99-107, 70-75).
    mode 0: A [M][K] B [N][K]^T;  1: A [M][K] B [K][N];  2: A [K][M]^T B [K][N].  The family is closed under differentiation
    (each gradient is another mode of the same Function), so gradients of every order exist, as for the convolutions."""

    @staticmethod
    def forward(ctx, a, b, mode, alpha=1.0):
        _ops._require_gpu(a)
        ctx.mode, ctx.alpha = mode, float(alpha)
        ctx.save_for_backward(a, b)
        ctx.out_dtype = torch.promote_types(a.dtype, b.dtype)      # the reference's addmm / matmul return the operands' dtype
        # fp64 operands: computed in fp32 on the HIP GEMM, RETURNED in fp64 as the reference's addmm / matmul would (the value carries
        # fp32 precision; callers that difference it numerically see a dtype-consistent graph)
        a, b = a.contiguous().float(), b.contiguous().float()
        if mode == 1 and a.shape[0] <= 32 and a.shape[1] >= 128:
            # a handful of rows (batch 16): the row-streaming kernel behind mode 0 (gemm_conv.hip: smallm_nt_kernel) wants K contiguous
            # in both operands -- transpose the (<= 16 MB) weight once instead of walking it with 4 workgroups
            b, mode = b.t().contiguous(), 0
        if mode == 0:
            (m, k), n = a.shape, b.shape[0]
            assert b.shape == (n, k)
        elif mode == 1:
            (m, k), n = a.shape, b.shape[1]
            assert b.shape == (k, n)
        else:
            (k, m), n = a.shape, b.shape[1]
            assert b.shape == (k, n)
        c = torch.empty(m, n, device=a.device, dtype=torch.float32)
        L.call("icg_gemm_batched", a, b, c, m, n, k, 1 if mode == 2 else 0, 1 if mode == 0 else 0, 0, 0, 0, 1, ctx.alpha)
        return c if ctx.out_dtype == torch.float32 else c.to(ctx.out_dtype)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        mode, al = ctx.mode, ctx.alpha
        da = db = None
        if mode == 0:        # C = al A B^T:  dA = al dC B,  dB = al dC^T A
            if ctx.needs_input_grad[0]:
                da = _Matmul.apply(dc, b, 1, al)
            if ctx.needs_input_grad[1]:
                db = _Matmul.apply(dc, a, 2, al)
        elif mode == 1:      # C = al A B:  dA = al dC B^T,  dB = al A^T dC
            if ctx.needs_input_grad[0]:
                da = _Matmul.apply(dc, b, 0, al)
            if ctx.needs_input_grad[1]:
                db = _Matmul.apply(a, dc, 2, al)
        else:                # C = al A^T B:  dA = al B dC^T,  dB = al A dC
            if ctx.needs_input_grad[0]:
                da = _Matmul.apply(b, dc, 0, al)
            if ctx.needs_input_grad[1]:
                db = _Matmul.apply(a, dc, 1, al)
        return da, db, None, None


def linear_nt(x, w, alpha=1.0):
    """alpha * x [M][K] @ w [N][K]^T -> [M][N] (F.linear without the bias), arbitrary-order gradients.  `alpha`: the equalised-
    learning-rate gain of a FullyConnectedLayer (networks.py:99-107 multiplies the weight by it first: one elementwise kernel
    forward and one backward per layer; here it rides in the GEMM's epilogue)."""
    return _Matmul.apply(x, w, 0, alpha)


def _one(v, what):
    if isinstance(v, (tuple, list)):
        if len(v) != 2 or v[0] != v[1]:
            raise NotImplementedError("%s must be the same in both dimensions (got %r)" % (what, v))
        v = v[0]
    return int(v)


def _check(input, weight, dilation, groups):
    assert isinstance(input, torch.Tensor) and input.ndim == 4 and weight.ndim == 4
    if _one(dilation, "dilation") != 1:
        raise NotImplementedError("conv2d_gradfix on HIP supports dilation=1 (got %r)" % (dilation,))
    if weight.shape[2] != weight.shape[3]:
        raise NotImplementedError("square kernels only")
    groups = int(groups)
    if groups < 1 or int(input.shape[1]) % groups or int(weight.shape[0]) % groups:
        raise ValueError("groups=%d does not divide the channel counts %d / %d" % (groups, input.shape[1], weight.shape[0]))


def _per_group(op, input, weight, bias, groups, **kw):
    """groups > 1 (the reference's only user: fused_modconv, one group per sample, networks.py:100-114): one call of the groups=1
    operator per group on channel slices, outputs concatenated -- F.conv2d's / F.conv_transpose2d's layout in both cases (group g
    owns input channels [g*Cin/G, (g+1)*Cin/G) and the weight's rows [g*W0/G, (g+1)*W0/G)).  Every order of gradient follows
    from the groups=1 operator's own."""
    xs, ws = input.chunk(groups, dim=1), weight.chunk(groups, dim=0)
    y = torch.cat([op(x, w, **kw) for x, w in zip(xs, ws)], dim=1)
    return _with_bias(y, bias)


def _with_bias(y, bias):
    return y if bias is None else y + bias.to(y.dtype).reshape(1, -1, 1, 1)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d semantics; weight [Cout][Cin][R][R]  (conv2d_gradfix.py:43-64)."""
    _check(input, weight, dilation, groups)
    if groups != 1:
        return _per_group(conv2d, input, weight, bias, int(groups), stride=stride, padding=padding)
    s, p, R = _one(stride, "stride"), _one(padding, "padding"), int(weight.shape[2])
    H, W = int(input.shape[2]), int(input.shape[3])
    out = ((H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1)
    if min(out) < 1 or p < 0:
        raise ValueError("conv2d: empty output / negative padding")
    geo = _Geo(R, s, p, 0, (H, W), out)
    cin = int(input.shape[1])
    if cin % 4 and cin >= 8:
        # channel counts like 513 (the discriminator epilogue's conv after MinibatchStd, training/networks.py:706-712): zero channels
        # up to a multiple of 4 keep the layer on the 16-byte-vector / split-K path (at 4x4 x batch 16 the scalar path is 8
        # workgroups running a 4 617-deep chain: 1 ms per launch); autograd slices the padding off again
        extra = 4 - cin % 4
        input = torch.nn.functional.pad(input, (0, 0, 0, 0, 0, extra))
        weight = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, extra))
    return _with_bias(_GatherConv.apply(input, weight.permute(0, 2, 3, 1), geo), bias)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    """F.conv_transpose2d semantics; weight [Cin][Cout][R][R]  (conv2d_gradfix.py:67-99)."""
    _check(input, weight, dilation, groups)
    if groups != 1:
        return _per_group(conv_transpose2d, input, weight, bias, int(groups), stride=stride, padding=padding,
                          output_padding=output_padding)
    s, p, op, R = _one(stride, "stride"), _one(padding, "padding"), _one(output_padding, "output_padding"), int(weight.shape[2])
    H, W = int(input.shape[2]), int(input.shape[3])
    out = ((H - 1) * s - 2 * p + R + op, (W - 1) * s - 2 * p + R + op)
    if min(out) < 1 or p < 0 or p > R - 1:
        raise ValueError("conv_transpose2d: unsupported padding / empty output")
    geo = _Geo(R, 1, R - 1 - p, s if s > 1 else 0, (H, W), out)
    w = weight.permute(1, 2, 3, 0).flip(1, 2)               # -> [Cout][R][R][Cin], taps flipped
    return _with_bias(_GatherConv.apply(input, w, geo), bias)

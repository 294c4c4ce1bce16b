"""StyleGAN2 layers as ONE autograd node each  ("style-scale / demod fused into conv").

The reference's training path executes a synthesis layer as ~12 framework operations forward and ~25 backward around its
convolution (stylegan2_ada_pytorch/training/networks.py:37-117 modulated_conv2d in its `fused_modconv=False` form, 405-444
SynthesisLayer.forward, 476-486 ToRGBLayer.forward, 226-242 Conv2dLayer.forward, 150-165 FullyConnectedLayer.forward); rounds 2 - 4
of this repository ran the same op graph with HIP kernels behind each operation and were bound by the host and by hundreds of
3 - 7 us launches.  Here a whole layer is one `torch.autograd.Function`:

    SynthesisLayer   modconv_layer    forward: affine GEMM, styles (+ fp16 pre-normalisation + demodulation coefficients), x * s,
                                      convolution [+ FIR], (x * d + noise + bias -> lrelu -> clamp)            = 5 - 6 launches
                                      backward: activation/demod/noise/bias gradients, [FIR^T], data and weight gradient of the
                                      convolution, modulation, style, affine and weight assembly               = 11 - 13 launches
    ToRGBLayer       torgb_layer      one pass over x forward (per-sample 3 x C weights, bias, clamp, image accumulation), one backward
    Conv2dLayer      conv_layer       [FIR] convolution (bias -> act -> clamp), weights prepared once per optimiser step
    FullyConnected   fc_layer         GEMM + (bias -> act); backward in two launches at the small batches of the mapping networks

with the weight-side work (gain / fp16 pre-normalisation, cast, gather layout, its adjoint, the demodulation table, Winograd / phase
forms) done ONCE per optimiser step for all layers of a network in two launches (`refresh`, icg_sg2_weight_prep_multi).

Two families of nodes.  `first_order()` (the Gmain and Dmain phases, every iteration; also `torch.no_grad()`: sampling, the generator
pass of Dmain): `once_differentiable` Functions whose backward is a sequence of kernel calls.  `second_order()` (the path-length
regulariser, every 4th iteration: loss.py:120-139 differentiates the generator's backward): `_ModConv2Fn` / `_ToRGB2Fn`, whose backward
is ITSELF a node (`_ModConvBwdFn` / `_ToRGBBwdFn`) with a hand-written adjoint -- affine / pre-normalisation, the style algebra, the
modulation (icg_sg2_mod2), the convolution K(u) with its weight gradient, the activation / demodulation block (icg_sg2_act_bwd2), K^T(cc),
and the demodulation's second-derivative term of the weight (icg_sg2_weight_bwd_q).  Those nodes are called with the tensors the layer
itself received (`x_in`, `wl_in`), never with re-laid-out copies, so that their cotangents reach the producer nodes.  R1 (every 16th
iteration) and any shape these kernels do not serve keep the composed, arbitrarily differentiable operators (stylegan_ops/modconv.py,
conv2d_resample.py, bias_act.py).
Same arithmetic as the composed path, including the places where fp16 tensors round (csrc/sg2_fused.hip)."""
import contextlib
import os
import weakref
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib as L
from .. import ops as _ops
from . import bias_act as _bias_act
from . import conv2d_gradfix as G
from . import upfirdn2d as U

ENABLED = os.environ.get("ICG_SG2_FUSED", "1") != "0"          # False: every layer keeps the composed operators (measurement / parity switch)
_FIRST_ORDER = 0


@contextlib.contextmanager
def first_order():
    """inside: the layers may assume that nobody differentiates their backward (Gmain / Dmain)"""
    global _FIRST_ORDER
    _FIRST_ORDER += 1
    try:
        yield
    finally:
        _FIRST_ORDER -= 1


_SECOND_ORDER = 0


@contextlib.contextmanager
def second_order():
    """inside: the modulated-convolution and toRGB layers are nodes whose backward is itself a node with a hand-written adjoint (the
    path-length regulariser differentiates the synthesis network's backward once more, loss.py:112-146); every other layer keeps its
    composed, arbitrarily differentiable operators"""
    global _SECOND_ORDER
    _SECOND_ORDER += 1
    try:
        yield
    finally:
        _SECOND_ORDER -= 1


SECOND_ORDER_ENABLED = os.environ.get("ICG_SG2_FUSED2", "1") != "0"      # False: the regulariser phases keep the composed operators


def twice():
    return _SECOND_ORDER > 0 and SECOND_ORDER_ENABLED


def active():
    """modulated-convolution / toRGB layers: fused in first-order phases, in second-order phases (twice-differentiable nodes), without autograd"""
    return ENABLED and (_FIRST_ORDER > 0 or twice() or not torch.is_grad_enabled())


def active1():
    """Conv2dLayer / FullyConnectedLayer: fused (once-differentiable) nodes in first-order phases and without autograd only"""
    return ENABLED and (_FIRST_ORDER > 0 or not torch.is_grad_enabled())


def _dt(t):
    return 1 if t.dtype == torch.float16 else 0


def _cl(x):
    return x.contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------------------------------ geometry
@dataclass(frozen=True)
class _Plan:
    """conv2d_resample.py:79-216 for the layer shapes of the networks: [FIR] -> gather convolution -> [FIR]"""
    pre: Optional[tuple]      # (up, down, (px0, px1, py0, py1), gain) of the FIR pass before the convolution
    geo: G._Geo
    post: Optional[tuple]
    flip: bool                # taps reversed in the gather weight


def _fir_out(n, up, down, p0, p1, fw):
    return (n * up + p0 + p1 - fw + down) // down


def plan(H, W, R, up, down, padding, fw, flip_weight):
    """None: a combination the fused layers leave to conv2d_resample"""
    if not isinstance(padding, int) or up not in (1, 2) or down not in (1, 2) or (up > 1 and down > 1) or R > 3:
        return None
    p0 = p1 = padding
    if up > 1:
        p0 += (fw + up - 1) // 2
        p1 += (fw - up) // 2
    if down > 1:
        p0 += (fw - down + 1) // 2
        p1 += (fw - down) // 2
    flip = not flip_weight
    if R == 1 and down > 1:
        h, w = _fir_out(H, 1, down, p0, p1, fw), _fir_out(W, 1, down, p0, p1, fw)
        return _Plan((1, down, (p0, p1, p0, p1), 1.0), G._Geo(1, 1, 0, 0, (h, w), (h, w)), None, flip)
    if R == 1 and up > 1:
        return None                                   # 1x1 + up (resnet generator skip): composed path
    if down > 1:
        h, w = _fir_out(H, 1, 1, p0, p1, fw), _fir_out(W, 1, 1, p0, p1, fw)
        out = ((h - R) // down + 1, (w - R) // down + 1)
        if min(h, w) < R:
            return None
        return _Plan((1, 1, (p0, p1, p0, p1), 1.0), G._Geo(R, down, 0, 0, (h, w), out), None, flip)
    if up > 1:
        p0 -= R - 1
        p1 -= R - up
        pt = max(min(-p0, -p1), 0)
        if pt > R - 1:
            return None
        out = ((H - 1) * up - 2 * pt + R, (W - 1) * up - 2 * pt + R)
        geo = G._Geo(R, 1, R - 1 - pt, up, (H, W), out)
        q0, q1 = p0 + pt, p1 + pt
        if _fir_out(out[0], 1, 1, q0, q1, fw) < 1:
            return None
        return _Plan(None, geo, (1, 1, (q0, q1, q0, q1), float(up ** 2)), flip)
    if p0 == p1 and p0 >= 0 and p0 <= R - 1:
        out = (H + 2 * p0 - R + 1, W + 2 * p0 - R + 1)
        if min(out) < 1:
            return None
        return _Plan(None, G._Geo(R, 1, p0, 0, (H, W), out), None, flip)
    return None


def _blur(x, f2, pad, flip, gain):
    """up = down = 1 FIR pass: the general upfirdn2d kernel (the strip kernel of csrc/sg2_fused.hip without its epilogue was measured and is not
    faster: 0.122 against 0.103 ms at [16, 64, 257, 257] fp16)"""
    return U._run(x, f2, (1, 1), (1, 1), pad, bool(flip), gain)


def _fir(x, f2, spec):
    up, down, pad, gain = spec
    if up == 1 and down == 1:
        return _blur(x, f2, pad, False, gain)
    return U._run(x, f2, (up, up), (down, down), pad, False, gain)


def _fir_act(x, f2, spec, d, noise, nbs, strength, bias, act, act_gain, clamp, keep_c):
    """the FIR pass `spec` (up = down = 1) with the layer epilogue on its results in one launch -> (c or None, y)"""
    up, down, (px0, px1, py0, py1), gain = spec
    assert up == 1 and down == 1
    x = _cl(x)
    N, C, H, W = (int(v) for v in x.shape)
    fh, fw = (int(v) for v in f2.shape)
    oh, ow = H + py0 + py1 - fh + 1, W + px0 + px1 - fw + 1
    y = torch.empty((N, C, oh, ow), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    c = torch.empty_like(y) if keep_c else None
    L.call("icg_sg2_fir_act_fwd", x, f2, c, y, d, noise, nbs, strength, bias, N, C, H, W, fh, fw, px0, px1, py0, py1, 0, float(gain), oh, ow, act,
           0.2, act_gain, clamp, _dt(x))
    return c, y


def _hconv_epilogue(x, I, O, pl):
    """does the fp16 convolution kernel take this layer with the epilogue on its accumulators (no FIR pass behind it)?"""
    return (x.dtype == torch.float16 and pl.post is None and pl.geo.zins == 0 and G.FP16_MFMA
            and bool(L.query("icg_conv2d_g_fprop_f16_applies", I, O, pl.geo.R, pl.geo.stride, 0)))


def _fir_adjoint(dy, f2, spec, in_hw):
    """upfirdn2d.py:329-346: the gradient of a FIR pass is the pass with up / down swapped, the filter flipped, adjoint padding"""
    up, down, (px0, px1, py0, py1), gain = spec
    ih, iw = in_hw
    oh, ow = int(dy.shape[2]), int(dy.shape[3])
    fh, fw = f2.shape
    p = (fw - px0 - 1, iw * up - ow * down + px0 - up + 1, fh - py0 - 1, ih * up - oh * down + py0 - up + 1)
    if up == 1 and down == 1:
        return _blur(dy, f2, p, True, gain)
    return U._run(dy, f2, (down, down), (up, up), p, True, gain)


# ------------------------------------------------------------------------------------------------------------------ prepared weights
class _Prep:
    """what a layer's kernels read instead of the parameter: valid while the parameter keeps its version and storage"""
    __slots__ = ("weight", "version", "ptr", "w_fwd", "w_adj", "wsq", "wscale", "warg", "prenorm", "gain", "flip", "cache")

    def stale(self):
        w = self.weight
        return self.version != w._version or self.ptr != w.data_ptr()

    def item(self):
        w = self.weight
        self.version, self.ptr = w._version, w.data_ptr()
        self.cache.clear()
        return dict(w=w.detach(), w_fwd=self.w_fwd, w_adj=self.w_adj, wsq=self.wsq, wscale=self.wscale, warg=self.warg,
                    prenorm=self.prenorm, gain=self.gain, flip=self.flip)


_PREPS = weakref.WeakKeyDictionary()      # layer module -> {configuration: _Prep}; not module state (nothing to pickle or deep-copy)


def _prep(owner, weight, dtype, prenorm, gain, flip, demod):
    store = _PREPS.get(owner)
    if store is None:
        store = _PREPS[owner] = {}
    key = (dtype, bool(prenorm), float(gain), bool(flip), bool(demod), weight.device)
    p = store.get(key)
    if p is None:
        O, I, R, _ = (int(v) for v in weight.shape)
        dev = weight.device
        p = _Prep()
        p.weight, p.prenorm, p.gain, p.flip, p.cache = weight, bool(prenorm), float(gain), bool(flip), {}
        p.w_fwd = torch.empty(O, R, R, I, device=dev, dtype=dtype)
        p.w_adj = torch.empty(I, R, R, O, device=dev, dtype=dtype)
        p.wsq = torch.empty(O, I, device=dev, dtype=torch.float32) if demod else None
        p.wscale = torch.empty(O, device=dev, dtype=torch.float32)
        p.warg = torch.empty(O, device=dev, dtype=torch.int32) if prenorm else None
        p.version, p.ptr = -1, 0
        store[key] = p
    if p.stale():
        _ops.sg2_weight_prep_multi([p.item()])
    return p


def refresh(*roots):
    """Re-prepare every stale prepared weight under the given modules in ONE icg_sg2_weight_prep_multi call (two launches per 40
    layers): called once per phase, after the optimiser has stepped.  Layers met for the first time prepare themselves lazily."""
    items = []
    for root in roots:
        if root is None:
            continue
        for m in root.modules():
            store = _PREPS.get(m)
            if store:
                for p in store.values():
                    if p.stale():
                        items.append(p.item())
    if items:
        _ops.sg2_weight_prep_multi(items)


def invalidate(*roots):
    """Drop every prepared weight under the given modules (all of them without arguments): the next forward of a layer prepares its
    weights again.  The prepared copies follow the parameter's version counter and storage pointer, which every in-place torch
    operation, optimiser step, `load_state_dict` and `.to()` moves; a write that moves neither -- through `p.data`, a raw pointer or
    a kernel of one's own -- must be followed by `ops.bump_version(p)` or by this call (the counterpart of
    `layers.invalidate_sn_cache` for the StyleGAN2 layers)."""
    if not roots:
        _PREPS.clear()
        return
    for root in roots:
        for m in root.modules():
            _PREPS.pop(m, None)


def _gemm_nt(a, b, alpha):
    """alpha a [M][K] b [N][K]^T on the HIP GEMM, fp32"""
    m, k = a.shape
    n = b.shape[0]
    c = torch.empty(m, n, device=a.device, dtype=torch.float32)
    L.call("icg_gemm_batched", a, b, c, m, n, k, 0, 1, 0, 0, 0, 1, float(alpha))
    return c


def _rows_ws(N, HW, C, ncols, dt, dev):
    nb = L.query("icg_sg2_rows_workspace_bytes", N, HW, C, ncols, dt)
    return _ops._bytes(nb, dev), nb


# ------------------------------------------------------------------------------------------------------------------ SynthesisLayer
@dataclass(frozen=True)
class ModConvCfg:
    plan: _Plan
    act_gain: float
    clamp: float              # < 0: none
    affine_wgain: float
    affine_bgain: float
    noise_bstride: int        # 0: one [H][W] noise image for the batch; H W: one per sample


def modconv_applies(x, weight, w_latent, affine_weight, up, padding, fw, flip_weight):
    if not (active() and (x.is_cuda or _EMULATED)) or x.dtype not in (torch.float32, torch.float16) or x.dim() != 4:
        return None
    O, I, R, R2 = (int(v) for v in weight.shape)
    dt = _dt(x)
    if R != R2 or R not in (1, 3) or int(x.shape[0]) > 64 or weight.dtype != torch.float32 or w_latent.dtype != torch.float32:
        return None
    if not (L.query("icg_sg2_rows_applies", I, dt) and L.query("icg_sg2_rows_applies", O, dt)):
        return None
    return plan(int(x.shape[2]), int(x.shape[3]), R, up, 1, padding, fw, flip_weight)


_EMULATED = False       # set by the CPU host-logic tests (kernels emulated by oracle/kernel_ref.py): lifts the is_cuda requirement


class _ModConvFn(Function):
    @staticmethod
    def forward(ctx, x, wl, aw, ab, weight, strength, bias, noise, f2, cfg, owner):
        x_in, wl_in = x, wl          # (the node's own inputs: the second-order form hands THEM to its backward node, not the re-laid-out copies)
        x = _cl(x)
        N, I, H, W = (int(v) for v in x.shape)
        O, R = int(weight.shape[0]), int(weight.shape[2])
        dt, dev, pl = _dt(x), x.device, cfg.plan
        half = dt == 1
        p = _prep(owner, weight, x.dtype, half, float(np.float32(1 / np.sqrt(I * R * R))) if half else 1.0, pl.flip, True)
        wl = wl.contiguous()
        lin = _gemm_nt(wl, aw.detach().contiguous(), cfg.affine_wgain)
        s = torch.empty(N, I, device=dev, dtype=torch.float32)
        d = torch.empty(N, O, device=dev, dtype=torch.float32)
        smax = torch.empty(N, device=dev, dtype=torch.float32) if half else None
        sarg = torch.empty(N, device=dev, dtype=torch.int32) if half else None
        L.call("icg_sg2_style_prep", lin, ab, cfg.affine_bgain, 1.0, p.wsq, N, I, O, int(half), s, smax, sarg, d)
        st = strength if noise is not None else None
        geo = pl.geo
        xs = None
        if half and G.FP16_MFMA and L.query("icg_modconv2d_f16_applies", I, O, R, geo.stride, geo.zins, geo.out[0], geo.out[1]):
            # ONE launch: x * s on the convolution's A fragments, the contraction, and -- without a blur behind it -- demodulation,
            # noise, bias, lrelu and clamp on its accumulators (csrc/hconv.hip, MOD / EP); x * s is never written
            c = torch.empty((N, O, geo.out[0], geo.out[1]), device=dev, dtype=x.dtype, memory_format=torch.channels_last)
            y = torch.empty_like(c) if pl.post is None else None
            L.call("icg_modconv2d_f16", x, s, p.w_fwd, c, y, d, noise, cfg.noise_bstride, st, bias, 3, 0.2, cfg.act_gain, cfg.clamp,
                   N, H, W, I, geo.out[0], geo.out[1], O, R, geo.stride, geo.pad, geo.zins)
            if pl.post is not None:      # up-sampling layer: the same epilogue on the blur's results
                c, y = _fir_act(c, f2, pl.post, d, noise, cfg.noise_bstride, st, bias, 3, cfg.act_gain, cfg.clamp, True)
        else:
            xs = torch.empty_like(x)
            L.call("icg_sg2_modulate", x, s, xs, N, H * W, I, dt)
            if _hconv_epilogue(x, I, O, pl):
                c = torch.empty((N, O, geo.out[0], geo.out[1]), device=dev, dtype=x.dtype, memory_format=torch.channels_last)
                y = torch.empty_like(c)
                L.call("icg_conv2d_g_fprop_f16_act", xs, p.w_fwd, c, y, d, noise, cfg.noise_bstride, st, bias, 3, 0.2, cfg.act_gain,
                       cfg.clamp, N, H, W, I, geo.out[0], geo.out[1], O, R, geo.stride, geo.pad)
            else:
                c = G.gather_conv(xs, p.w_fwd, geo, p.cache)
                if pl.post is not None:
                    c, y = _fir_act(c, f2, pl.post, d, noise, cfg.noise_bstride, st, bias, 3, cfg.act_gain, cfg.clamp, True)
                else:
                    y = torch.empty_like(c)
                    L.call("icg_sg2_act_fwd", c, d, noise, cfg.noise_bstride, st, bias, y, N, int(c.shape[2]) * int(c.shape[3]), O, 3,
                           0.2, cfg.act_gain, cfg.clamp, dt)
        Ho, Wo = int(c.shape[2]), int(c.shape[3])
        ctx.cfg, ctx.p, ctx.dims = cfg, p, (N, I, H, W, O, R, Ho, Wo)
        ctx.save_for_backward(x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2, ab, x_in, wl_in)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return _modconv_backward(ctx.saved_tensors[:13], ctx.cfg, ctx.p, ctx.dims, ctx.needs_input_grad, dy, None) + (None, None, None, None)


def _modconv_backward(saved, cfg, p, dims, needs, dy, keep):
    """first-order gradients of a modulated-convolution layer as kernel calls -> (dx, dwl, daw, dab, dweight, dstrength, dbias);
    `keep` (a dict or None) receives the intermediates the second-order pass needs"""
    x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2 = saved
    N, I, H, W, O, R, Ho, Wo = dims
    pl, dt, dev = cfg.plan, _dt(x), x.device
    dy = _cl(dy.to(y.dtype))
    dc = torch.empty_like(dy)
    sums = torch.empty(N, 2 * O + 1, device=dev, dtype=torch.float32)
    tot = torch.empty(2 * O + 1, device=dev, dtype=torch.float32)
    ws, nb = _rows_ws(N, Ho * Wo, O, 2 * O + 1, dt, dev)
    L.call("icg_sg2_act_bwd", dy, y, c, d, noise, cfg.noise_bstride, dc, sums, tot, N, Ho * Wo, O, 3, 0.2, cfg.act_gain, cfg.clamp, dt,
           ws, nb)
    if pl.post is not None:
        dc = _fir_adjoint(dc, f2, pl.post, pl.geo.out)
    need_w = needs[4]
    need_s = any(needs[1:4])
    dx = dwl = daw = dab = dweight = None
    t = dxs = g = pdot = None
    if needs[0] or need_s:
        dxs = G.gather_conv(dc, p.w_adj, pl.geo.adjoint(), p.cache)
        dxo = torch.empty_like(x) if needs[0] else None
        ds = torch.empty(N, I, device=dev, dtype=torch.float32)
        ws, nb = _rows_ws(N, H * W, I, I, dt, dev)
        L.call("icg_sg2_modulate_bwd", dxs, x, s, dxo, ds, N, H * W, I, dt, ws, nb)
        dx = dxo
    if need_s or need_w:
        nblk = (I + 63) // 64
        g = torch.empty(N, I, device=dev, dtype=torch.float32)
        pdot = torch.empty(N, nblk, device=dev, dtype=torch.float32)
        t = torch.empty(N, O, device=dev, dtype=torch.float32)
        if not (needs[0] or need_s):
            ds = torch.zeros(N, I, device=dev, dtype=torch.float32)
        L.call("icg_sg2_style_bwd", ds, I, sums[:, O:], 2 * O + 1, d, s, p.wsq, N, I, O, g, pdot, t)
        if need_s:
            K = int(wl.shape[1])
            daw = torch.empty_like(aw) if needs[2] else None
            dab = torch.empty(I, device=dev, dtype=torch.float32) if needs[3] else None
            dwl = torch.empty_like(wl) if needs[1] else None
            L.call("icg_sg2_fc_bwd", g, smax, sarg, pdot if smax is not None else None, nblk, 1.0, wl, aw, N, I, K, cfg.affine_wgain,
                   cfg.affine_bgain, daw, dab, dwl)
    if need_w and not G.weight_gradients_disabled:
        if xs is None:      # the forward modulated inside the convolution: the weight gradient's operand is rebuilt here
            xs = torch.empty_like(x)
            L.call("icg_sg2_modulate", x, s, xs, N, H * W, I, dt)
        tw, layout = G.gather_wgrad_raw(xs, dc, pl.geo)
        dweight = torch.empty_like(weight)
        nbw = L.query("icg_sg2_weight_bwd_workspace_bytes", O, I) if p.prenorm else 0
        L.call("icg_sg2_weight_bwd", tw, layout, t, s, N, weight, p.wscale, p.warg, int(p.prenorm), p.gain, dt, dweight, O, I, R,
               _ops._bytes(nbw, dev) if nbw else None, nbw)
    dstrength = tot[2 * O].reshape(()) if (noise is not None and needs[5]) else None
    dbias = tot[:O] if needs[6] else None
    if keep is not None:
        keep.update(dy=dy, dc_conv=dc, dxs=dxs, g=g, pdot=pdot, t=t, dd=sums[:, O:2 * O].contiguous(), xs=xs)
    return dx, dwl, daw, dab, dweight, dstrength, dbias


def _mm(a, b, ta=False, tb=False, alpha=1.0):
    """alpha op(a) op(b) on the HIP GEMM (fp32, small matrices of the style / affine algebra): no vendor library in the step"""
    a, b = a.contiguous(), b.contiguous()
    m = a.shape[1] if ta else a.shape[0]
    k = a.shape[0] if ta else a.shape[1]
    n = b.shape[0] if tb else b.shape[1]
    c = torch.empty(m, n, device=a.device, dtype=torch.float32)
    L.call("icg_gemm_batched", a, b, c, int(m), int(n), int(k), int(ta), int(tb), 0, 0, 0, 1, float(alpha))
    return c


class _ModConv2Fn(Function):
    """the same layer for phases that differentiate TWICE: its backward is the node _ModConvBwdFn, whose adjoint is written out below"""

    @staticmethod
    def forward(ctx, x, wl, aw, ab, weight, strength, bias, noise, f2, cfg, owner):
        return _ModConvFn.forward(ctx, x, wl, aw, ab, weight, strength, bias, noise, f2, cfg, owner)

    @staticmethod
    def backward(ctx, dy):
        x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2, ab, x_in, wl_in = ctx.saved_tensors
        st = dict(saved=(x.detach(), xs, c, y, s, d, smax, sarg, wl.detach(), aw.detach(), weight.detach(), noise, f2), cfg=ctx.cfg, p=ctx.p,
                  dims=ctx.dims, needs=ctx.needs_input_grad)
        out = _ModConvBwdFn.apply(dy, x_in, wl_in, aw, ab, weight, st)
        return tuple(out) + (None, None, None, None)


class _ModConvBwdFn(Function):
    """B(dy; x, wl, A, ab, W) -> (dx, dwl, dA, dab, dW, dstrength, dbias): the first-order gradients as a node.  Its backward -- the adjoint of B
    with respect to (dy, x, wl, A, W) for cotangents of dx and dwl, what the path-length penalty sends back -- follows B's own steps in reverse and
    then the layer's forward dependencies (c = K(x s, W~), d = (s^2 wsq + eps)^-1/2, s = N(lin), lin = wg wl A^T + ab):
        affine / pre-normalisation  ->  style algebra on [N x I], [N x O] matrices (ATen, no autograd)  ->  u = x cot(ds) + cot(dx) s
        -> K(u), wgrad(u, dc)  ->  icg_sg2_act_bwd2  ->  K^T(cot c), wgrad(x s, cot c)  ->  icg_sg2_modulate_bwd  ->  the same algebra backwards"""

    @staticmethod
    def forward(ctx, dy, x, wl, aw, ab, weight, st):
        keep = {}
        outs = _modconv_backward(st["saved"], st["cfg"], st["p"], st["dims"], st["needs"], dy, keep)
        ctx.st, ctx.keep = st, keep
        ctx.mark_non_differentiable(*[o for o in outs[2:] if o is not None])
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, cdx, cdwl, *rest):
        st, kp = ctx.st, ctx.keep
        x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2 = st["saved"]
        cfg, p, (N, I, H, W, O, R, Ho, Wo) = st["cfg"], st["p"], st["dims"]
        pl, dt, dev = cfg.plan, _dt(x), x.device
        half = smax is not None
        wg, bg = cfg.affine_wgain, cfg.affine_bgain
        dy, dc_conv, dxs, g, t, dd = kp["dy"], kp["dc_conv"], kp["dxs"], kp["g"], kp["t"], kp["dd"]
        wsq = p.wsq
        ar = torch.arange(N, device=dev)
        f32 = dict(device=dev, dtype=torch.float32)
        # ---- affine layer and pre-normalisation of B:  dwl = wg dlin A,  dlin = J^T g
        cl = _mm(cdwl.float(), aw, False, True, wg) if cdwl is not None else torch.zeros(N, I, **f32)          # cot(dlin) = wg cdwl A^T
        if half:
            m, sg, e = smax.abs(), torch.sign(smax), sarg.long()
            P = (g * s).sum(1)
            dlin = g / m[:, None]
            dlin[ar, e] -= sg * P / m
            cle = cl[ar, e]
            cg = cl / m[:, None] - s * (sg * cle / m)[:, None]
            cot_m = -(cl * g).sum(1) / (m * m) + cle * sg * P / (m * m)
            cs = -(cle * sg / m)[:, None] * g
        else:
            dlin, cg, cs, cot_m = g, cl, torch.zeros(N, I, **f32), None
        c_aw = _mm(dlin, cdwl.float(), True, False, wg) if cdwl is not None else torch.zeros_like(aw)          # cot(A) = wg dlin^T cdwl
        # ---- g = ds_mod + s (t wsq),  t = -dd d^3
        tw_ = _mm(t, wsq)
        cs = cs + cg * tw_
        cgs = cg * s
        cot_t = _mm(cgs, wsq, False, True)
        cwsq = _mm(t, cgs, True, False)
        d3 = d * d * d
        cdd = -cot_t * d3
        cd = -3.0 * cot_t * dd * d * d
        # ---- ds_mod = sum_p dxs x,  dx = dxs s:  u = cot(dxs)
        s_r = s.half().float() if dt == 1 else s
        u = torch.empty_like(x)
        L.call("icg_sg2_mod2", x, cg, cdx if cdx is None else _cl(cdx.to(x.dtype)), s_r if cdx is not None else None, u, N, H * W, I, dt)
        cx1 = torch.empty_like(x)
        if cdx is not None:
            sums1 = torch.empty(N, I, **f32)
            ws, nb = _rows_ws(N, H * W, I, I, dt, dev)
            L.call("icg_sg2_modulate_bwd", dxs, _cl(cdx.to(x.dtype)), cg, cx1, sums1, N, H * W, I, dt, ws, nb)        # dxs cot(ds), sum_p dxs cdx
            cs = cs + sums1
        else:
            L.call("icg_sg2_modulate", dxs, cg, cx1, N, H * W, I, dt)
        # ---- dxs = K^T(dc, W~):  cot(dc) = K(u),  cot(W~) += wgrad(u, dc)
        cdc = G.gather_conv(u, p.w_fwd, pl.geo, p.cache)
        if pl.post is not None:
            cdc = _fir(cdc, f2, pl.post)
        tw1, layout = G.gather_wgrad_raw(u, dc_conv, pl.geo)
        # ---- dc = dz d, dd = sum_p dz c, dz = dy m
        cdy = torch.empty_like(dy)
        cc = torch.empty_like(dy)
        sums2 = torch.empty(N, O, **f32)
        ws, nb = _rows_ws(N, Ho * Wo, O, O, dt, dev)
        L.call("icg_sg2_act_bwd2", dy, y, c, cdc, d, cdd.contiguous(), cdy, cc, sums2, N, Ho * Wo, O, 3, 0.2, cfg.act_gain, cfg.clamp, dt, ws, nb)
        cd = cd + sums2
        # ---- the layer's forward dependencies: c = K(xs, W~), xs = x s
        cc_conv = _fir_adjoint(cc, f2, pl.post, pl.geo.out) if pl.post is not None else cc
        cxs = G.gather_conv(cc_conv, p.w_adj, pl.geo.adjoint(), p.cache)
        if xs is None:
            xs = kp.get("xs")
        if xs is None:
            xs = torch.empty_like(x)
            L.call("icg_sg2_modulate", x, s, xs, N, H * W, I, dt)
        tw2, layout2 = G.gather_wgrad_raw(xs, cc_conv, pl.geo)
        assert layout2 == layout
        cx2 = torch.empty_like(x)
        sums3 = torch.empty(N, I, **f32)
        ws, nb = _rows_ws(N, H * W, I, I, dt, dev)
        L.call("icg_sg2_modulate_bwd", cxs, x, s, cx2, sums3, N, H * W, I, dt, ws, nb)
        cs = cs + sums3
        cx = cx1 + cx2
        # ---- d = (q + eps)^-1/2, q = s^2 wsq^T
        cq = -0.5 * d3 * cd
        cs = cs + 2.0 * s * _mm(cq, wsq)
        cwsq = cwsq + _mm(cq, s * s, True, False)
        # ---- s = N(lin), lin = wg wl A^T + bg ab
        if half:
            clin = cs / m[:, None]
            clin[ar, e] -= sg * (cs * s).sum(1) / m
            clin[ar, e] += sg * cot_m
        else:
            clin = cs
        c_wl = _mm(clin, aw, False, False, wg)
        c_aw = c_aw + _mm(clin, wl, True, False, wg)
        c_ab = clin.sum(0) * bg
        # ---- W~ = W scale(W), wsq = sum_k W~^2
        cweight = torch.empty_like(weight)
        nbw = L.query("icg_sg2_weight_bwd_workspace_bytes", O, I) if p.prenorm else 0
        L.call("icg_sg2_weight_bwd_q", tw1 + tw2, layout, None, None, 0, (2.0 * cwsq).contiguous(), weight, p.wscale, p.warg, int(p.prenorm), p.gain, 0,
               cweight, O, I, R, _ops._bytes(nbw, dev) if nbw else None, nbw)
        return cdy, cx, c_wl, c_aw, c_ab, cweight, None


def modconv_layer(owner, x, w_latent, affine, weight, strength, bias, noise, noise_bstride, f2, pl, act_gain, clamp):
    cfg = ModConvCfg(pl, float(act_gain), float(clamp if clamp is not None else -1), float(affine.weight_gain), float(affine.bias_gain),
                     int(noise_bstride))
    fn = _ModConv2Fn if (twice() and torch.is_grad_enabled()) else _ModConvFn
    return fn.apply(x, w_latent, affine.weight, affine.bias, weight, strength, bias, noise, f2, cfg, owner)


# ------------------------------------------------------------------------------------------------------------------ ToRGBLayer
def torgb_applies(x, weight, w_latent):
    if not (active() and (x.is_cuda or _EMULATED)) or x.dtype not in (torch.float32, torch.float16) or x.dim() != 4:
        return False
    return (int(weight.shape[0]) == 3 and int(weight.shape[2]) == 1 and int(x.shape[0]) <= 64 and w_latent.dtype == torch.float32
            and bool(L.query("icg_sg2_torgb_applies", int(weight.shape[1]), _dt(x))))


class _ToRGBFn(Function):
    @staticmethod
    def forward(ctx, x, wl, aw, ab, weight, bias, img, wgain, awgain, abgain, clamp):
        x_in, wl_in = x, wl
        x = _cl(x)
        N, C, H, W = (int(v) for v in x.shape)
        dt, dev = _dt(x), x.device
        wl = wl.contiguous()
        lin = _gemm_nt(wl, aw.detach().contiguous(), awgain)
        s = torch.empty(N, C, device=dev, dtype=torch.float32)
        L.call("icg_sg2_style_prep", lin, ab, abgain, wgain, None, N, C, 0, 0, s, None, None, None)
        y = torch.empty(N, H * W, 3, device=dev, dtype=x.dtype)
        if img is not None:
            img = img.contiguous()
            assert img.dtype == torch.float32 and tuple(img.shape) == (N, 3, H, W)
        out = torch.empty(N, 3, H, W, device=dev, dtype=torch.float32)
        L.call("icg_sg2_torgb_fwd", x, s, weight.detach().contiguous(), bias, clamp, img, out, y, N, H * W, C, dt)
        ctx.k = (wgain, awgain, abgain, clamp, N, C, H, W, img is not None)
        ctx.save_for_backward(x, y, s, wl, aw, weight, ab, x_in, wl_in)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dimg):
        return _torgb_backward(ctx.saved_tensors[:6], ctx.k, ctx.needs_input_grad, dimg, None) + (None, None, None, None)


def _torgb_backward(saved, k, needs, dimg, keep):
    x, y, s, wl, aw, weight = saved
    wgain, awgain, abgain, clamp, N, C, H, W, has_img = k
    dt, dev = _dt(x), x.device
    dimg = dimg.contiguous()
    need_s = any(needs[1:4])
    dx = torch.empty_like(x) if needs[0] else None
    ctot = 4 * C + 3
    sums = torch.empty(N, ctot, device=dev, dtype=torch.float32)
    tot = torch.empty(ctot, device=dev, dtype=torch.float32)
    nb = L.query("icg_sg2_torgb_bwd_workspace_bytes", N, H * W, C, dt)
    L.call("icg_sg2_torgb_bwd", dimg, y, x, s, weight.detach().contiguous(), clamp, int(not _bias_act.REFERENCE_CLAMP_GRAD), dx, sums,
           tot, N, H * W, C, dt, _ops._bytes(nb, dev), nb)
    dwl = daw = dab = None
    if need_s:
        nblk = (C + 63) // 64
        g = torch.empty(N, C, device=dev, dtype=torch.float32)
        pdot = torch.empty(N, nblk, device=dev, dtype=torch.float32)
        L.call("icg_sg2_style_bwd", sums, ctot, None, 0, None, s, None, N, C, 0, g, pdot, None)
        K = int(wl.shape[1])
        daw = torch.empty_like(aw) if needs[2] else None
        dab = torch.empty(C, device=dev, dtype=torch.float32) if needs[3] else None
        dwl = torch.empty_like(wl) if needs[1] else None
        L.call("icg_sg2_fc_bwd", g, None, None, None, 0, wgain, wl, aw, N, C, K, awgain, abgain, daw, dab, dwl)
    dweight = tot[C:4 * C].reshape(weight.shape) if needs[4] else None
    dbias = tot[4 * C:] if needs[5] else None
    if keep is not None:
        keep.update(dimg=dimg, ds=sums[:, :C].contiguous())
    return dx, dwl, daw, dab, dweight, dbias, (dimg if has_img and needs[6] else None)


class _ToRGB2Fn(Function):
    """ToRGB for phases that differentiate twice: its backward is the node _ToRGBBwdFn"""

    @staticmethod
    def forward(ctx, x, wl, aw, ab, weight, bias, img, wgain, awgain, abgain, clamp):
        return _ToRGBFn.forward(ctx, x, wl, aw, ab, weight, bias, img, wgain, awgain, abgain, clamp)

    @staticmethod
    def backward(ctx, dimg):
        x, y, s, wl, aw, weight, ab, x_in, wl_in = ctx.saved_tensors
        st = dict(saved=(x.detach(), y, s, wl.detach(), aw.detach(), weight.detach()), k=ctx.k, needs=ctx.needs_input_grad)
        out = _ToRGBBwdFn.apply(dimg, x_in, wl_in, aw, ab, weight, st)
        return tuple(out) + (None, None, None, None)


class _ToRGBBwdFn(Function):
    """B(dimg; x, wl, A, ab, W) -> (dx, dwl, dA, dab, dW, dbias, dimg_in); adjoint for cotangents of dx, dwl and dimg_in (icg_sg2_torgb_bwd2)"""

    @staticmethod
    def forward(ctx, dimg, x, wl, aw, ab, weight, st):
        keep = {}
        outs = _torgb_backward(st["saved"], st["k"], st["needs"], dimg, keep)
        ctx.st, ctx.keep = st, keep
        ctx.mark_non_differentiable(*[o for o in outs[2:6] if o is not None])
        if outs[6] is not None:           # (the image gradient passes through: a view of the incoming tensor would alias an input of this node)
            outs = outs[:6] + (outs[6].clone(),)
        return outs

    @staticmethod
    @once_differentiable
    def backward(ctx, cdx, cdwl, c2, c3, c4, c5, cim):
        st, kp = ctx.st, ctx.keep
        x, y, s, wl, aw, weight = st["saved"]
        wgain, awgain, abgain, clamp, N, C, H, W, has_img = st["k"]
        dt, dev = _dt(x), x.device
        ds = kp["ds"]
        dlin = ds * wgain
        if cdwl is not None:
            cl = _mm(cdwl.float(), aw, False, True, awgain)
            c_aw = _mm(dlin, cdwl.float(), True, False, awgain)
        else:
            cl, c_aw = torch.zeros(N, C, device=dev), torch.zeros_like(aw)
        a = (cl * wgain).contiguous()
        cdimg = torch.empty(N, 3, H, W, device=dev, dtype=torch.float32)
        cx = torch.empty_like(x)
        sums = torch.empty(N, 4 * C, device=dev, dtype=torch.float32)
        tot = torch.empty(4 * C, device=dev, dtype=torch.float32)
        nb = L.query("icg_sg2_torgb_bwd_workspace_bytes", N, H * W, C, dt)
        L.call("icg_sg2_torgb_bwd2", kp["dimg"], y, x, s, weight.contiguous(), a, None if cdx is None else _cl(cdx.to(x.dtype)),
               None if cim is None else cim.contiguous(), clamp, int(not _bias_act.REFERENCE_CLAMP_GRAD), cdimg, cx, sums, tot, N, H * W, C, dt,
               _ops._bytes(nb, dev), nb)
        clin = sums[:, :C] * wgain
        c_wl = _mm(clin, aw, False, False, awgain)
        c_aw = c_aw + _mm(clin, wl, True, False, awgain)
        c_ab = clin.sum(0) * abgain
        return cdimg, cx, c_wl, c_aw, c_ab, tot[C:4 * C].reshape(weight.shape).clone(), None


def torgb_layer(x, w_latent, affine, weight, bias, img, weight_gain, clamp):
    """-> img + torgb(x)   (img may be None), fp32 NCHW"""
    fn = _ToRGB2Fn if (twice() and torch.is_grad_enabled()) else _ToRGBFn
    return fn.apply(x, w_latent, affine.weight, affine.bias, weight, bias, img, float(weight_gain), float(affine.weight_gain),
                    float(affine.bias_gain), float(clamp if clamp is not None else -1))


# ------------------------------------------------------------------------------------------------------------------ Conv2dLayer
_ACT_IDS = {"linear": 1, "lrelu": 3}


def conv_applies(x, weight, activation, up, down, padding, fw, flip_weight):
    if not (active1() and (x.is_cuda or _EMULATED)) or x.dtype not in (torch.float32, torch.float16) or x.dim() != 4:
        return None
    O, I, R, R2 = (int(v) for v in weight.shape)
    if R != R2 or R not in (1, 3) or activation not in _ACT_IDS or weight.dtype != torch.float32 or (I % 4 and I >= 8):
        return None          # (I = 513, the epilogue's convolution after MinibatchStd: conv2d_gradfix.conv2d pads the channels)
    if not L.query("icg_sg2_rows_applies", O, _dt(x)):
        return None
    return plan(int(x.shape[2]), int(x.shape[3]), R, up, down, padding, fw, flip_weight)


@dataclass(frozen=True)
class ConvCfg:
    plan: _Plan
    act: int
    weight_gain: float
    act_gain: float
    clamp: float
    has_bias: bool


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, f2, cfg, owner):
        N = int(x.shape[0])
        O, I, R = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        dt, pl = _dt(x), cfg.plan
        ctx.from_rgb = False
        if (I == 3 and R == 1 and pl.pre is None and pl.post is None and pl.geo.stride == 1 and pl.geo.pad == 0
                and (cfg.has_bias or cfg.act != 1 or cfg.clamp >= 0) and L.query("icg_sg2_fromrgb_applies", O, dt)):
            # fromRGB: y written once from the planar image (csrc/sg2_fused.hip) instead of an fp32 [pixels x 3] GEMM, a cast and the
            # activation pass
            xin = x.contiguous()
            H, W = int(x.shape[2]), int(x.shape[3])
            p = _prep(owner, weight, x.dtype, False, cfg.weight_gain, pl.flip, False)
            y = torch.empty((N, O, H, W), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
            L.call("icg_sg2_fromrgb_fwd", xin, p.w_fwd, bias, y, N, H * W, O, cfg.act, 0.2, cfg.act_gain, cfg.clamp, dt)
            ctx.cfg, ctx.p, ctx.plain, ctx.in_hw, ctx.from_rgb = cfg, p, False, (H, W), True
            ctx.save_for_backward(xin, y, weight, f2)
            return y
        x = _cl(x)
        # a pure gain (no bias, linear, no clamp: the resnet skip) rides in the prepared weight
        plain = cfg.act == 1 and not cfg.has_bias and cfg.clamp < 0
        p = _prep(owner, weight, x.dtype, False, cfg.weight_gain * (cfg.act_gain if plain else 1.0), pl.flip, False)
        xin = _fir(x, f2, pl.pre) if pl.pre is not None else x
        if not plain and _hconv_epilogue(xin, I, O, pl):
            Ho, Wo = pl.geo.out
            y = torch.empty((N, O, Ho, Wo), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
            L.call("icg_conv2d_g_fprop_f16_act", xin, p.w_fwd, None, y, None, None, 0, None, bias, cfg.act, 0.2, cfg.act_gain, cfg.clamp,
                   N, int(xin.shape[2]), int(xin.shape[3]), I, Ho, Wo, O, R, pl.geo.stride, pl.geo.pad)
        else:
            c = G.gather_conv(xin, p.w_fwd, pl.geo, p.cache)
            if pl.post is not None:
                c = _fir(c, f2, pl.post)
            y = c
            if not plain:
                y = torch.empty_like(c)
                L.call("icg_sg2_act_fwd", c, None, None, 0, None, bias, y, N, int(c.shape[2]) * int(c.shape[3]), O, cfg.act, 0.2, cfg.act_gain,
                       cfg.clamp, dt)
        ctx.cfg, ctx.p, ctx.plain, ctx.in_hw = cfg, p, plain, (int(x.shape[2]), int(x.shape[3]))
        ctx.save_for_backward(xin, None if plain else y, weight, f2)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        xin, y, weight, f2 = ctx.saved_tensors
        cfg, p, pl = ctx.cfg, ctx.p, ctx.cfg.plan
        N = int(xin.shape[0])
        O, I, R = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        dt, dev = _dt(xin), xin.device
        dz = _cl(dy.to(xin.dtype))
        if ctx.from_rgb:
            HW = ctx.in_hw[0] * ctx.in_hw[1]
            tot = torch.empty(O, 4, device=dev, dtype=torch.float32)
            dimg = torch.empty_like(xin) if ctx.needs_input_grad[0] else None
            ws, nb = _rows_ws(N, HW, O, 4 * O, dt, dev)
            L.call("icg_sg2_fromrgb_bwd", dz, y, xin, p.w_fwd, dimg, tot, N, HW, O, cfg.act, 0.2, cfg.act_gain, cfg.clamp, dt, ws, nb)
            dweight = (tot[:, :3] * p.gain).reshape(weight.shape) if (ctx.needs_input_grad[1] and not G.weight_gradients_disabled) else None
            dbias = tot[:, 3].contiguous() if (cfg.has_bias and ctx.needs_input_grad[2]) else None
            return dimg, dweight, dbias, None, None, None
        dbias = None
        if not ctx.plain:
            HWo = int(dz.shape[2]) * int(dz.shape[3])
            tot = torch.empty(2 * O + 1, device=dev, dtype=torch.float32)
            out = torch.empty_like(dz)
            ws, nb = _rows_ws(N, HWo, O, 2 * O + 1, dt, dev)
            L.call("icg_sg2_act_bwd", dz, y, None, None, None, 0, out, None, tot, N, HWo, O, cfg.act, 0.2, cfg.act_gain, cfg.clamp, dt, ws, nb)
            dz = out
            if cfg.has_bias and ctx.needs_input_grad[2]:
                dbias = tot[:O]
        if pl.post is not None:
            dz = _fir_adjoint(dz, f2, pl.post, pl.geo.out)
        dx = dweight = None
        if ctx.needs_input_grad[0]:
            dx = G.gather_conv(dz, p.w_adj, pl.geo.adjoint(), p.cache)
            if pl.pre is not None:
                dx = _fir_adjoint(dx, f2, pl.pre, ctx.in_hw)
        if ctx.needs_input_grad[1] and not G.weight_gradients_disabled:
            tw, layout = G.gather_wgrad_raw(xin, dz, pl.geo)
            dweight = torch.empty_like(weight)
            L.call("icg_sg2_weight_bwd", tw, layout, None, None, 0, weight, p.wscale, None, 0, 1.0, dt, dweight, O, I, R, None, 0)
        return dx, dweight, dbias, None, None, None


def conv_layer(owner, x, weight, bias, f2, pl, activation, weight_gain, act_gain, clamp):
    cfg = ConvCfg(pl, _ACT_IDS[activation], float(weight_gain), float(act_gain), float(clamp if clamp is not None else -1), bias is not None)
    return _ConvFn.apply(x, weight, bias, f2, cfg, owner)


# ------------------------------------------------------------------------------------------------------------------ FullyConnectedLayer
def fc_applies(x, weight, activation):
    return bool(active1() and (x.is_cuda or _EMULATED) and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
                and int(x.shape[0]) <= 64 and activation in _ACT_IDS)


class _FCFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, wgain, bgain, act_gain):
        x = x.contiguous()
        lin = _gemm_nt(x, weight.detach().contiguous(), wgain)
        y = lin
        if bias is not None or act != 1:
            y = torch.empty_like(lin)
            if bias is not None and bgain != 1:
                bias = bias.detach() * bgain
            L.call("icg_bias_act", lin, bias, None, None, None, y, lin.numel(), 1, int(lin.shape[1]) if bias is not None else 1, 0, act, 0.2,
                   act_gain, -1.0)
        ctx.k = (act, wgain, bgain, act_gain)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y if act != 1 else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        act, wgain, bgain, act_gain = ctx.k
        dy = dy.contiguous()
        N, K = (int(v) for v in x.shape)
        I = int(weight.shape[0])
        dz = dy
        if act != 1 or act_gain != 1:
            dz = torch.empty_like(dy)
            L.call("icg_bias_act", dy, None, None, y, None, dz, dy.numel(), 1, 1, 1, act, 0.2, act_gain, -1.0)
        dW = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(I, device=x.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if dW is not None or db is not None or dx is not None:
            L.call("icg_sg2_fc_bwd", dz, None, None, None, 0, 1.0, x, weight, N, I, K, wgain, bgain, dW, db, dx)
        return dx, dW, db, None, None, None, None


def fc_layer(x, weight, bias, activation, weight_gain, bias_gain, act_gain):
    return _FCFn.apply(x, weight, bias, _ACT_IDS[activation], float(weight_gain), float(bias_gain), float(act_gain))

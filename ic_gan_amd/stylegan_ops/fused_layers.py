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

The Functions are first order only (`once_differentiable`): `loss.accumulate_gradients` enters `first_order()` for the Gmain and
Dmain phases, which run every iteration; the lazy regularisers (path length every 4th, R1 every 16th iteration) differentiate twice
and keep the composed operators (stylegan_ops/modconv.py, conv2d_resample.py, bias_act.py), as does any shape these kernels do not
serve.  Without autograd (`torch.no_grad()`: sampling, the generator pass of Dmain) the fused forward is used as well.
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


def active():
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
        ctx.save_for_backward(x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, xs, c, y, s, d, smax, sarg, wl, aw, weight, noise, f2 = ctx.saved_tensors
        cfg, p, (N, I, H, W, O, R, Ho, Wo) = ctx.cfg, ctx.p, ctx.dims
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
        need_w = ctx.needs_input_grad[4]
        need_s = any(ctx.needs_input_grad[1:4])
        dx = dwl = daw = dab = dweight = None
        t = None
        if ctx.needs_input_grad[0] or need_s:
            dxs = G.gather_conv(dc, p.w_adj, pl.geo.adjoint(), p.cache)
            dxo = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            ds = torch.empty(N, I, device=dev, dtype=torch.float32)
            ws, nb = _rows_ws(N, H * W, I, I, dt, dev)
            L.call("icg_sg2_modulate_bwd", dxs, x, s, dxo, ds, N, H * W, I, dt, ws, nb)
            dx = dxo
        if need_s or need_w:
            nblk = (I + 63) // 64
            g = torch.empty(N, I, device=dev, dtype=torch.float32)
            pdot = torch.empty(N, nblk, device=dev, dtype=torch.float32)
            t = torch.empty(N, O, device=dev, dtype=torch.float32)
            if not (ctx.needs_input_grad[0] or need_s):
                ds = torch.zeros(N, I, device=dev, dtype=torch.float32)
            L.call("icg_sg2_style_bwd", ds, I, sums[:, O:], 2 * O + 1, d, s, p.wsq, N, I, O, g, pdot, t)
            if need_s:
                K = int(wl.shape[1])
                daw = torch.empty_like(aw) if ctx.needs_input_grad[2] else None
                dab = torch.empty(I, device=dev, dtype=torch.float32) if ctx.needs_input_grad[3] else None
                dwl = torch.empty_like(wl) if ctx.needs_input_grad[1] else None
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
        dstrength = tot[2 * O].reshape(()) if (noise is not None and ctx.needs_input_grad[5]) else None
        dbias = tot[:O] if ctx.needs_input_grad[6] else None
        return dx, dwl, daw, dab, dweight, dstrength, dbias, None, None, None, None


def modconv_layer(owner, x, w_latent, affine, weight, strength, bias, noise, noise_bstride, f2, pl, act_gain, clamp):
    cfg = ModConvCfg(pl, float(act_gain), float(clamp if clamp is not None else -1), float(affine.weight_gain), float(affine.bias_gain),
                     int(noise_bstride))
    return _ModConvFn.apply(x, w_latent, affine.weight, affine.bias, weight, strength, bias, noise, f2, cfg, owner)


# ------------------------------------------------------------------------------------------------------------------ ToRGBLayer
def torgb_applies(x, weight, w_latent):
    if not (active() and (x.is_cuda or _EMULATED)) or x.dtype not in (torch.float32, torch.float16) or x.dim() != 4:
        return False
    return (int(weight.shape[0]) == 3 and int(weight.shape[2]) == 1 and int(x.shape[0]) <= 64 and w_latent.dtype == torch.float32
            and bool(L.query("icg_sg2_torgb_applies", int(weight.shape[1]), _dt(x))))


class _ToRGBFn(Function):
    @staticmethod
    def forward(ctx, x, wl, aw, ab, weight, bias, img, wgain, awgain, abgain, clamp):
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
        ctx.save_for_backward(x, y, s, wl, aw, weight)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dimg):
        x, y, s, wl, aw, weight = ctx.saved_tensors
        wgain, awgain, abgain, clamp, N, C, H, W, has_img = ctx.k
        dt, dev = _dt(x), x.device
        dimg = dimg.contiguous()
        need_s = any(ctx.needs_input_grad[1:4])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
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
            daw = torch.empty_like(aw) if ctx.needs_input_grad[2] else None
            dab = torch.empty(C, device=dev, dtype=torch.float32) if ctx.needs_input_grad[3] else None
            dwl = torch.empty_like(wl) if ctx.needs_input_grad[1] else None
            L.call("icg_sg2_fc_bwd", g, None, None, None, 0, wgain, wl, aw, N, C, K, awgain, abgain, daw, dab, dwl)
        dweight = tot[C:4 * C].reshape(weight.shape) if ctx.needs_input_grad[4] else None
        dbias = tot[4 * C:] if ctx.needs_input_grad[5] else None
        return dx, dwl, daw, dab, dweight, dbias, (dimg if has_img and ctx.needs_input_grad[6] else None), None, None, None, None


def torgb_layer(x, w_latent, affine, weight, bias, img, weight_gain, clamp):
    """-> img + torgb(x)   (img may be None), fp32 NCHW"""
    return _ToRGBFn.apply(x, w_latent, affine.weight, affine.bias, weight, bias, img, float(weight_gain), float(affine.weight_gain),
                          float(affine.bias_gain), float(clamp if clamp is not None else -1))


# ------------------------------------------------------------------------------------------------------------------ Conv2dLayer
_ACT_IDS = {"linear": 1, "lrelu": 3}


def conv_applies(x, weight, activation, up, down, padding, fw, flip_weight):
    if not (active() and (x.is_cuda or _EMULATED)) or x.dtype not in (torch.float32, torch.float16) or x.dim() != 4:
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
    return bool(active() and (x.is_cuda or _EMULATED) and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
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

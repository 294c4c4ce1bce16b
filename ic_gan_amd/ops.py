"""Autograd operators of the IC-GAN hot path, each a thin shim over the C-ABI of
libicgan_hip.so (include/icgan_hip.h).  PyTorch supplies device memory, the current
HIP stream and the autograd tape; all arithmetic on activations and weights runs in the
hand-written gfx950 kernels.  No operator has a CPU / eager fallback.

Layout: activations are logical NCHW tensors stored channels-last (NHWC memory), which
is what the kernels address; `_cl` makes that true at module boundaries.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, replace
from typing import Optional

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import _lib as L


def _cl(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 4:
        raise ValueError("expected a 4-D activation")
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous(memory_format=torch.channels_last)


def _empty_cl(b, c, h, w, dev):
    return torch.empty((b, c, h, w), device=dev, dtype=torch.float32, memory_format=torch.channels_last)


def _bytes(n, dev):
    return torch.empty(max(int(n), 16), device=dev, dtype=torch.uint8)


def _f32(n, dev):
    return torch.empty(int(n), device=dev, dtype=torch.float32)


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("ic_gan_amd operators run on an AMD GPU only (tensor is on %s); there is no CPU path"
                           % t.device)


# ----------------------------------------------------------------------------------------------
# spectral norm state  (reference: layers.py:39-61, 98-112)
# ----------------------------------------------------------------------------------------------
@dataclass
class SNState:
    w_ohwi: torch.Tensor            # W/sigma, [Cout][R][R][Cin]
    w_dgrad: Optional[torch.Tensor]  # W/sigma, [Cin][R][R][Cout], taps flipped
    u: torch.Tensor                 # u' used for sigma (saved copy)
    v: torch.Tensor
    sigma: torch.Tensor
    rows: int
    cin: int
    R: int
    w_up: Optional[torch.Tensor] = None      # phase weights of the upsample-fused conv, [4][Cout][2][2][Cin]
    w_up_dgrad: Optional[torch.Tensor] = None  # [Cin][4][4][Cout]
    w_down: Optional[torch.Tensor] = None    # 4x4/stride-2 kernel of conv3x3 -> avgpool2, [Cout][4][4][Cin]
    w_down_dgrad: Optional[torch.Tensor] = None  # [4][Cin][2][2][Cout]
    w_wino: Optional[torch.Tensor] = None        # Winograd-domain weight [16][Cout][Cin] (wide 3x3 stride-1 layers)
    w_wino_dgrad: Optional[torch.Tensor] = None  # the same for the data gradient, [16][Cin][Cout]
    wino_m: int = 0                              # 2: F(2x2,3x3), 16 planes;  4: F(4x4,3x3), 36 planes;
    #                                              5: resample-fused layer in the 25-plane F(4x4,3x3) domain, per direction:
    rs: tuple = (False, False, False)            # (fprop, dgrad, wgrad) run in that domain (else phase / 4x4-stride-2 form)
    raw_numel: int = 0                           # elements of the RAW weight gradient this layer's backward produces (HWIO / OHWI:
    #                                              rows cin R R; phase form dw_up / pooled form dw_down: 16 rows cin)
    handle: Optional[torch.Tensor] = None        # grouped spectral-norm backward (SNGroupFn): the tensor that stands for the weight
    #                                              in the layer's autograd node; its gradient is the raw weight gradient
    group_forms: Optional[tuple] = None          # (list shared with the SNGroupFn node, this layer's index): which argument of
    #                                              icg_sn_backward the gradient is (0 hwio, 1 ohwi, 2 up, 3 down), written in backward


def _sn_alloc(weight, need_dgrad, upsample, downsample, winograd=False):
    """-> (contiguous weight, SNState with its output buffers, scratch buffer)"""
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    rows = w.shape[0]
    if w.dim() == 4:
        cin, R = w.shape[1], w.shape[2]
        assert w.shape[2] == w.shape[3]
    else:
        cin, R = w[0].numel(), 1
    dev = w.device
    n = rows * cin * R * R
    up = bool(upsample) and R == 3
    down = bool(downsample) and R == 3
    rs = (False, False, False)
    if (up or down) and int(winograd) == 5:
        rs = resample_winograd_directions(cin, rows, up)
    st = SNState(_f32(n, dev), _f32(n, dev) if (need_dgrad and ((not up and not down) or rs[1])) else None, _f32(rows, dev),
                 _f32(cin * R * R, dev), _f32(1, dev), rows, cin, R)
    st.raw_numel = 16 * rows * cin if ((up or down) and not rs[2]) else n
    if any(rs):
        st.wino_m, st.rs = 5, rs
        st.w_wino = _f32(25 * rows * cin, dev) if rs[0] else None
        st.w_wino_dgrad = _f32(25 * rows * cin, dev) if (need_dgrad and rs[1]) else None
    if up:
        st.w_up = _f32(16 * rows * cin, dev) if not rs[0] else None
        st.w_up_dgrad = _f32(16 * rows * cin, dev) if (need_dgrad and not rs[1]) else None
    if down:
        st.w_down = _f32(16 * rows * cin, dev) if not rs[0] else None
        st.w_down_dgrad = _f32(16 * rows * cin, dev) if (need_dgrad and not rs[1]) else None
    if winograd and R == 3 and not up and not down:
        st.wino_m = 4 if int(winograd) == 4 else 2
        planes = 36 if st.wino_m == 4 else 16
        st.w_wino = _f32(planes * rows * cin, dev)
        st.w_wino_dgrad = _f32(planes * rows * cin, dev) if need_dgrad else None
    nb = L.query("icg_sn_scratch_bytes", rows, cin, R)
    return w, st, _bytes(nb, dev)


def _sn_winograd(st: SNState):
    """Winograd-domain copies of W/sigma (after the spectral-norm pass filled w_ohwi / w_dgrad)."""
    fn = {5: "icg_wino4r_weight_transform", 4: "icg_wino4_weight_transform"}.get(st.wino_m, "icg_wino_weight_transform")
    if st.w_wino is not None:
        L.call(fn, st.w_ohwi, st.w_wino, st.rows, st.cin)
    if st.w_wino_dgrad is not None:
        L.call(fn, st.w_dgrad, st.w_wino_dgrad, st.cin, st.rows)


def _sn_winograd_many(states):
    """`_sn_winograd` of many layers as ONE launch (icg_wino_weight_transform_multi; ~80 tiny launches per step otherwise)."""
    import ctypes
    jobs = []
    for st in states:
        planes = {5: 25, 4: 36}.get(st.wino_m, 16)
        if st.w_wino is not None:
            jobs.append((st.w_ohwi, st.w_wino, st.rows, st.cin, planes))
        if st.w_wino_dgrad is not None:
            jobs.append((st.w_dgrad, st.w_wino_dgrad, st.cin, st.rows, planes))
    if not jobs:
        return
    arr = (L.WinoWeight * len(jobs))()
    for i, (w, U, n, k, planes) in enumerate(jobs):
        arr[i].w, arr[i].U, arr[i].N, arr[i].K, arr[i].planes = w.data_ptr(), U.data_ptr(), n, k, planes
    L.call("icg_wino_weight_transform_multi", ctypes.cast(arr, ctypes.c_void_p), len(jobs))


def sn_prepare(weight: torch.Tensor, u: torch.Tensor, sv: Optional[torch.Tensor], eps: float, training: bool,
               need_dgrad: bool, upsample: bool = False, downsample: bool = False, winograd: bool = False) -> SNState:
    """One power iteration (updates `u`/`sv` in place when training) and W/sigma in kernel layouts.
    upsample=True (3x3 conv that follows a nearest x2 upsample) emits the 4-phase 2x2 layouts instead of OHWI-dgrad."""
    _require_gpu(weight)
    w, st, scratch = _sn_alloc(weight, need_dgrad, upsample, downsample, winograd)
    L.call("icg_sn_forward", w, u, sv, st.rows, st.cin, st.R, float(eps), int(bool(training)), st.v, st.u, st.sigma,
           st.w_ohwi, st.w_dgrad, st.w_up, st.w_up_dgrad, st.w_down, st.w_down_dgrad, scratch, scratch.numel())
    if training:
        bump_version(u, sv)
    _sn_winograd(st)
    return st


def sn_prepare_many(items, eps: float, training: bool):
    """`sn_prepare` for many layers in one batched pass (icg_sn_forward_multi): items = [(weight, u, sv, need_dgrad,
    upsample, downsample[, winograd]), ...] -> [SNState, ...].  Bit-identical to calling sn_prepare per layer."""
    import ctypes
    if not items:
        return []
    _require_gpu(items[0][0])
    arr = (L.SnLayer * len(items))()
    states, keep = [], []
    for i, (weight, u, sv, need_dgrad, upsample, downsample, *rest) in enumerate(items):
        w, st, scratch = _sn_alloc(weight, need_dgrad, upsample, downsample, int(rest[0]) if rest else 0)
        keep.append((w, scratch))
        d = arr[i]
        d.w, d.u, d.sv = w.data_ptr(), u.data_ptr(), (sv.data_ptr() if sv is not None else None)
        d.v_out, d.u_out, d.sigma_out = st.v.data_ptr(), st.u.data_ptr(), st.sigma.data_ptr()
        d.w_ohwi = st.w_ohwi.data_ptr()
        for name, t in (("w_dgrad", st.w_dgrad), ("w_up_fprop", st.w_up), ("w_up_dgrad", st.w_up_dgrad),
                        ("w_down_fprop", st.w_down), ("w_down_dgrad", st.w_down_dgrad)):
            setattr(d, name, t.data_ptr() if t is not None else None)
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel()
        d.rows, d.Cin, d.R = st.rows, st.cin, st.R
        states.append(st)
    L.call("icg_sn_forward_multi", ctypes.cast(arr, ctypes.c_void_p), len(items), float(eps), int(bool(training)))
    if training:
        for it in items:
            bump_version(it[1], it[2])
    _sn_winograd_many(states)
    return states


SN_BACKWARD_GROUP = int(os.environ.get("ICG_SN_BWD_GROUP", "8"))   # layers per SNGroupFn node; 0: spectral-norm backward per layer


def _sn_backward(dw_hwio, dw_ohwi, sn: SNState, like: torch.Tensor, dw_up=None, dw_down=None) -> torch.Tensor:
    """Gradient of the PARAMETER from the raw gradient of W / sigma (autograd of SN.W_, reference layers.py:98-112).  When the layer's
    autograd node was given the group handle in place of the weight (`like is sn.handle`), the raw gradient itself is returned: it
    travels to SNGroupFn, which runs this for the whole group in two launches."""
    if sn.handle is not None and like is sn.handle:
        forms = (dw_hwio, dw_ohwi, dw_up, dw_down)
        given = [i for i, t in enumerate(forms) if t is not None]
        if len(given) != 1 or forms[given[0]].numel() != sn.raw_numel:
            raise RuntimeError("grouped spectral-norm backward: expected one raw weight gradient of %d elements" % sn.raw_numel)
        sn.group_forms[0][sn.group_forms[1]] = given[0]
        return forms[given[0]].reshape(-1)
    dw = torch.empty_like(like, memory_format=torch.contiguous_format)
    nb = L.query("icg_sn_backward_scratch_bytes", sn.rows, sn.cin, sn.R)
    scratch = _bytes(nb, like.device)
    L.call("icg_sn_backward", dw_hwio, dw_ohwi, dw_up, dw_down, sn.w_ohwi, sn.u, sn.v, sn.sigma, sn.rows, sn.cin, sn.R, dw, 0,
           scratch, nb)
    return dw


def sn_backward_many(items):
    """`_sn_backward` of many layers in two launches per 16 (icg_sn_backward_multi): items = [(raw gradient, form, SNState, like)]
    -> [dweight, ...]; bit-identical to the per-layer calls."""
    import ctypes
    arr = (L.SnBwdItem * len(items))()
    out, keep = [], []
    for i, (raw, form, sn, like) in enumerate(items):
        raw = raw.contiguous()
        dw = torch.empty_like(like, memory_format=torch.contiguous_format)
        nb = L.query("icg_sn_backward_scratch_bytes", sn.rows, sn.cin, sn.R)
        scratch = _bytes(nb, like.device)
        d = arr[i]
        for k, name in enumerate(("dw_hwio", "dw_ohwi", "dw_up", "dw_down")):
            setattr(d, name, raw.data_ptr() if k == form else None)
        d.w_ohwi, d.u, d.v, d.sigma = sn.w_ohwi.data_ptr(), sn.u.data_ptr(), sn.v.data_ptr(), sn.sigma.data_ptr()
        d.dw, d.scratch, d.scratch_bytes = dw.data_ptr(), scratch.data_ptr(), nb
        d.rows, d.Cin, d.R, d.accumulate = sn.rows, sn.cin, sn.R, 0
        out.append(dw)
        keep.append((raw, scratch))
    L.call("icg_sn_backward_multi", ctypes.cast(arr, ctypes.c_void_p), len(items))
    return out


class SNGroupFn(Function):
    """One autograd node in front of a GROUP of spectrally normalised layers: inputs = their weight parameters, outputs = one
    HANDLE per layer (a storage-free tensor with the shape of that layer's raw weight gradient) which the layer's own node takes
    in place of the weight.  The layers' backward passes return raw gradients of W / sigma; autograd delivers them here once the
    whole group has run, and the spectral-norm backward (reference: autograd through SN.W_, layers.py:98-112) of all of them is two
    launches instead of two per layer.  Groups are consecutive layers (SN_BACKWARD_GROUP), so under DDP the parameters of a group
    become ready together and the bucketed all-reduce still overlaps the rest of the backward pass."""

    @staticmethod
    def forward(ctx, states, *weights):
        # The node must not reference the SNStates: they hold the handles (= this node's outputs), and a cycle node -> state ->
        # handle -> node keeps every step's saved activations alive until Python's cycle collector happens to run (GPU memory is
        # invisible to it: the cfg3 bench ran out of 288 GB in 25 steps).  It keeps copies without the handle, and the list the
        # layers' backward passes write their gradient form into.
        ctx.forms = [-1] * len(states)
        ctx.states = tuple(replace(st, handle=None, group_forms=None) for st in states)
        ctx.likes = weights
        for i, st in enumerate(states):
            st.group_forms = (ctx.forms, i)
        ctx.set_materialize_grads(False)
        return tuple(w.new_empty(1).expand(st.raw_numel) for st, w in zip(states, weights))

    @staticmethod
    def backward(ctx, *grads):
        items, where = [], []
        for i, (st, g, like) in enumerate(zip(ctx.states, grads, ctx.likes)):
            if g is not None and ctx.needs_input_grad[1 + i]:
                if ctx.forms[i] < 0:
                    raise RuntimeError("grouped spectral-norm backward: a gradient arrived for a handle no layer consumed")
                items.append((g, ctx.forms[i], st, like))
                where.append(i)
        out = [None] * len(grads)
        if items:
            for i, dw in zip(where, sn_backward_many(items)):
                out[i] = dw
        return (None,) + tuple(out)


def sn_group(states, weights):
    """Give consecutive groups of `states` (prepared for `weights`, one forward of a network under autograd) their SNGroupFn
    handles.  No-op without autograd or when no weight requires a gradient."""
    if SN_BACKWARD_GROUP <= 0 or not torch.is_grad_enabled():
        return
    for i in range(0, len(states), SN_BACKWARD_GROUP):
        st, ws = states[i:i + SN_BACKWARD_GROUP], weights[i:i + SN_BACKWARD_GROUP]
        if not any(w.requires_grad for w in ws):
            continue
        for s, h in zip(st, SNGroupFn.apply(tuple(st), *ws)):
            s.handle = h if h.requires_grad else None


def _conv_fprop(x, w, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, R, flags):
    """icg_conv2d_fprop, with the split-K workspace when the launch would not fill the chip (small batch / low resolution)."""
    nb = L.query("icg_conv2d_fprop_workspace_bytes", B, H, W, Cin, Cout, R, flags)
    if nb:
        L.call("icg_conv2d_fprop_ws", x, w, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, R, flags, 1.0,
               _bytes(nb, out.device), nb)
    else:
        L.call("icg_conv2d_fprop", x, w, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, R, flags, 1.0)


FUSE_RELU_BACKWARD = True       # ReLU backward of a ReLU-only prologue in the data-gradient epilogue (ICG_RES_RELU_MASK)
# A layer's weight gradient (transform of dy: HBM-bound; plane / split-K GEMMs: MFMA-bound; spectral-norm backward) depends on
# the forward's saved tensors and dout only, its data gradient (transform, GEMM, transform) on dout and the weights only: the
# two are launched on two HIP streams and overlap inside every layer's backward -- the MFMA-bound phases of one run beside the
# HBM-bound phases of the other instead of after them.  The main stream waits for the layer's weight gradient before the
# backward node returns, so everything downstream (autograd accumulation, DDP's bucket hooks, Adam) sees finished gradients.
WGRAD_SIDE_STREAM = False     # measured (profiles/r03_wgrad_side_stream.txt): 264.4 -> 263.0 ms per step only -- both kernels fill the chip, the
#                               second one's workgroups are dispatched as the first one's drain -- and the per-kernel timings stop being
#                               comparable with a profile; kept as an opt-in (bench.py --wgrad-stream), validated by the whole GPU suite
_WGRAD_STREAMS = {}


def _wgrad_stream(dev):
    dev = torch.device(dev)
    if dev.type != "cuda":
        return None
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _WGRAD_STREAMS.get(key)
    if st is None:
        st = _WGRAD_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st

KEEP_WINOGRAD_V = True          # keep the forward pass's transformed input for the weight gradient (memory for one HBM pass)


def _wgrad_from_v(v, dout, dw_hwio, B, H, W, Cin, Cout, planes, dy_up, dy_alpha, want_dbias):
    """Weight gradient from the forward pass's V planes; when the layer's bias gradient is wanted as well it comes out of the
    same pass over dy (icg_conv2d_wino4_wgrad_from_v_db) instead of a separate icg_colsum.  -> dbias or None"""
    dev = dout.device
    if want_dbias and Cout <= 2048:
        dbias = _f32(Cout, dev)
        nb = L.query("icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes", B, H, W, Cin, Cout, planes)
        L.call("icg_conv2d_wino4_wgrad_from_v_db", v, dout, dw_hwio, dbias, B, H, W, Cin, Cout, planes, dy_up, dy_alpha,
               _bytes(nb, dev), nb)
        return dbias
    nb = L.query("icg_conv2d_wino4_wgrad_from_v_workspace_bytes", B, H, W, Cin, Cout, planes)
    L.call("icg_conv2d_wino4_wgrad_from_v", v, dout, dw_hwio, B, H, W, Cin, Cout, planes, dy_up, dy_alpha, _bytes(nb, dev), nb)
    return None


def _saved_v(ws, planes, B, H, W, Cin):
    """the V = transform(act(x)) region the F(4x4,3x3) forward entries leave at the start of their workspace (H, W: full
    resolution); the view keeps the workspace alive until the backward pass has used it"""
    return ws[: planes * B * (H // 4) * (W // 4) * Cin * 4].view(torch.float32)


def _wino_fprop(x, U, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, flags, m=2, keep_v=False):
    v = "wino4" if m == 4 else "wino"
    nb = L.query("icg_conv2d_%s_workspace_bytes" % v, B, H, W, Cin, Cout)
    ws = _bytes(nb, out.device)
    keep = bool(keep_v and m == 4 and KEEP_WINOGRAD_V)
    # ICG_WINO_KEEP_V: V is read back below (the fused narrow-layer kernel writes it only on request, csrc/fwino.hip)
    L.call("icg_conv2d_%s_fprop" % v, x, U, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout,
           flags | (L.ICG_WINO_KEEP_V if keep else 0), 1.0, ws, nb)
    return _saved_v(ws, 36, B, H, W, Cin) if keep else None


WINOGRAD_WGRAD = True            # weight gradient of those layers through the Winograd domain as well
# measured on MI355X (tools/wino_bench.py -> profiles/r01_wino_microbench.txt, speed-up over the direct implicit GEMM, B = 64):
#   F(2x2,3x3): 96 ch 0.8x, 192 ch 1.2x, 384 ch 1.6x, 768 ch 1.9x, 1536 ch 2.8x
#   F(4x4,3x3): 96 ch 1.3x, 192 ch 1.9x, 384 ch 2.4x, 768 ch 3.2x, 1536 ch 4.2x   (weight gradient: 1.3 / 1.8 / 2.6 / 3.0 / 3.2x)
WINOGRAD_MIN_CHANNELS = 96       # from here a Winograd form: F(4x4,3x3) when H, W are multiples of 4 ...
WINOGRAD4_MIN_CHANNELS = 96
WINOGRAD2_MIN_CHANNELS = 192     # ... else F(2x2,3x3), which needs 192 channels to pay for its 4x-volume transforms


def winograd_applies(cin, cout, h, w, batch):
    """0, or the Winograd output-tile size (2 / 4) to use for a 3x3 / stride-1 layer of this shape."""
    c = min(cin, cout)
    if c < WINOGRAD_MIN_CHANNELS or cin % 4 or cout % 4 or h % 2 or w % 2 or 4 * batch * h * w >= 0x7FFFFFFF:
        return 0
    if h % 4 == 0 and w % 4 == 0 and c >= max(WINOGRAD4_MIN_CHANNELS, WINOGRAD_MIN_CHANNELS):
        return 4
    return 2 if c >= WINOGRAD2_MIN_CHANNELS else 0


# resample-fused layers in the 25-plane domain (tools/rs_wino_bench.py, speed-up over the phase / 4x4-stride-2 forms at
# min(Cin, Cout) = 96 / 192 / 384 / 768 / 1536):  upsample-fused fprop 1.1 / 1.6 / 2.0 / 2.1 / 2.6x, dgrad 0.9 / 1.3 / 1.8 / 2.0 / 4.8x,
# wgrad 1.2 / 1.9 / 2.1 / 2.2 / 2.0x;  pool-fused fprop 0.7 / 1.1 / 1.6 / 1.9 / 2.2x, dgrad 0.9 / 1.3 / 1.7 / 2.1 / 2.3x, wgrad 0.7 / 1.05 / 1.6 / 1.8 / 2.1x
# round 3, against the second-generation kernels (profiles/r03_rs_winograd_thresholds.txt): upsample-fused 192 -> 96 @256: fprop 1.39x, dgrad 1.13x,
# wgrad 1.42x; pool-fused 96 -> 96 @256: 0.97 / 1.05 / 0.98x (stays on the 4x4-stride-2 form), 192 -> 192 @128: 1.49 / 1.51 / 1.37x
# round 4: with the fused narrow-layer kernel (csrc/fwino.hip) behind the same entry points the pool-fused 96 -> 96 @256 layer runs
# 1.26x faster in the 25-plane domain than on the 4x4-stride-2 form (profiles/r04_fwino_microbench.txt): forward and data gradient
# from 96 channels; its weight gradient stays on the direct kernel (no V planes are kept for it)
RS_WINOGRAD_MIN_CHANNELS = {True: (96, 96, 96), False: (96, 96, int(os.environ.get("ICG_RS_POOL_WGRAD_MIN", "192")))}         # upsample?: (fprop, dgrad, wgrad)


def fwino_applies(B, H, W, Cin, Cout) -> bool:
    """does the fused F(4x4,3x3) kernel take this layer (H, W: its full resolution)?  (icg_fwino_applies, csrc/fwino.hip)"""
    return bool(L.query("icg_fwino_applies", B, H, W, Cin, Cout))


def resample_winograd_applies(cin, cout, h, w, batch):
    """5 when a resample-fused 3x3 layer (h, w: its FULL resolution) can run in the 25-plane Winograd domain, else 0.
    Deliberately independent of the batch size (up to the index-range check): the layouts a layer's spectral-norm pass
    emits are prefetched from its previous call (layers.sn_prefetch), and D alternates between batch 2B and B."""
    if cin % 4 or cout % 4 or h % 4 or w % 4 or 36 * batch * (h // 4) * (w // 4) >= 0x7FFFFFFF:
        return 0
    return 5 if any(resample_winograd_directions(cin, cout, True)) or any(resample_winograd_directions(cin, cout, False)) else 0


def resample_winograd_directions(cin, cout, upsample):
    """(fprop, dgrad, wgrad) in the 25-plane domain?  The pool-fused forward / data gradient go there from 96 channels only because the
    fused narrow-layer kernel (csrc/fwino.hip) takes them; below 192 channels the three-kernel composite it would otherwise fall to is
    slower than the 4x4-stride-2 form, so that range is admitted only for widths the fused kernel serves (Cin % 32 == 0 and a column
    count of 96 k) -- a batch-independent proxy of icg_fwino_applies, which keeps the prefetched spectral-norm layouts stable
    (ADVICE r04)."""
    c = min(cin, cout)
    th = RS_WINOGRAD_MIN_CHANNELS[bool(upsample)]
    if not upsample and c < 192 and not (cin % 32 == 0 and cout % 96 == 0 and cin % 96 == 0):
        th = tuple(max(t, 192) for t in th)
    return tuple(c >= m for m in th)


WINOGRAD4_WGRAD_MIN_CHANNELS = 96       # weight gradient: F(4x4,3x3) domain from here (2.25x transform volume instead of 4x)


def disable_winograd():
    """Route every 3x3 layer through the implicit-GEMM / 2x2-phase / 4x4-stride-2 kernels only (no Winograd transforms):
    the arithmetic then differs from a direct convolution by summation order alone.  `bench.py --no-winograd`."""
    global WINOGRAD_MIN_CHANNELS, WINOGRAD2_MIN_CHANNELS, WINOGRAD4_MIN_CHANNELS, WINOGRAD4_WGRAD_MIN_CHANNELS
    global RS_WINOGRAD_MIN_CHANNELS
    big = 10 ** 9
    WINOGRAD_MIN_CHANNELS = WINOGRAD2_MIN_CHANNELS = WINOGRAD4_MIN_CHANNELS = WINOGRAD4_WGRAD_MIN_CHANNELS = big
    RS_WINOGRAD_MIN_CHANNELS = {True: (big, big, big), False: (big, big, big)}


def winograd_wgrad_tile(cin, cout, h, w, batch):
    """0 (direct weight gradient), or the Winograd tile size (2 / 4) for the weight gradient of a 3x3 / stride-1 layer."""
    if not WINOGRAD_WGRAD or not winograd_applies(cin, cout, h, w, batch):
        return 0
    return 4 if (h % 4 == 0 and w % 4 == 0 and min(cin, cout) >= WINOGRAD4_WGRAD_MIN_CHANNELS) else 2


# ----------------------------------------------------------------------------------------------
# fused [BN/ccbn apply + ReLU + nearest-upsample] -> conv / linear -> [+bias +residual]
# (reference: layers.py:144-153, 164-165, 398-437, 485-503, 542-552, 587-613)
# ----------------------------------------------------------------------------------------------
@dataclass
class BNOpt:
    running_mean: torch.Tensor
    running_var: torch.Tensor
    eps: float
    momentum: float
    training: bool
    gain_offset: float          # 1.0 for ccbn (1 + gain(y)), 0.0 for plain bn (gain is the parameter)
    sync_group: object = None   # torch.distributed process group (or True for WORLD) -> cross-replica statistics


@dataclass
class ConvOpt:
    sn: SNState
    relu: bool = False
    upsample: bool = False
    res_up: bool = False
    bn: Optional[BNOpt] = None
    downsample: bool = False    # conv3x3 -> 2x2 average pool, run as one 4x4/stride-2 conv (needs sn.w_down)
    bn_stats: object = None     # BNStats started ahead by bn_stats_begin (cross-replica BN: all-reduce in flight)
    chain: bool = False         # also return the input as a second output whose gradient is folded into this layer's data gradient


@dataclass
class BNStats:
    """Training-mode batch statistics of one BN input, possibly still being all-reduced across replicas."""
    x: torch.Tensor              # the channels-last tensor the statistics were taken of
    sums: torch.Tensor           # double[2C] about `shift`, or the packed cross-replica payload double[2C+1] about 0
    shift: Optional[torch.Tensor]
    count: float                 # element count per channel; 0.0 = on the device, sums[2C] (cross-replica)
    work: object = None          # pending torch.distributed work of the all-reduce

    def wait(self):
        if self.work is not None:
            self.work.wait()     # RCCL: the current stream waits for the collective's stream; gloo: host wait
            self.work = None


def _sync_enabled(bn: Optional[BNOpt]) -> bool:
    return bn is not None and bn.sync_group is not None and dist.is_available() and dist.is_initialized() \
        and dist.get_world_size(None if bn.sync_group is True else bn.sync_group) > 1


def _group(bn: BNOpt):
    return None if bn.sync_group is True else bn.sync_group


class FusedConvFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, gain, beta, opt: ConvOpt):
        _require_gpu(x)
        x_in = x
        x = _cl(x)
        if opt.chain and (x.data_ptr() != x_in.data_ptr() or x.dtype != x_in.dtype):
            x_in = x                          # a layout copy was made: chain on the copy
        sn = opt.sn
        B, Cin, Hs, Ws = x.shape
        assert Cin == sn.cin, (Cin, sn.cin)
        up = 1 if opt.upsample else 0
        H, W = Hs << up, Ws << up
        Cout, R = sn.rows, sn.R
        down = bool(opt.downsample)
        if down:
            assert (sn.w_down is not None or sn.rs[0]) and opt.bn is None and not up and Hs % 2 == 0 and Ws % 2 == 0
            H, W = Hs // 2, Ws // 2
        dev = x.device
        flags = (L.ICG_PRE_RELU if opt.relu else 0) | (L.ICG_UPSAMPLE2X if up else 0)
        scale = shift = mean = invstd = None
        ssb = 0
        count = float(B * Hs * Ws)
        bn = opt.bn
        gb_rows = 1
        count_dev = None
        if bn is not None:
            if opt.bn_stats is not None:
                # the statistics must have been started on THIS input: same channels-last storage (the usual case: _cl was a
                # no-op both times) or, when a layout copy was made on either side, the same logical tensor
                sx = opt.bn_stats.x
                if sx.shape != x.shape or (sx.data_ptr() != x.data_ptr() and not torch.equal(sx, x)):
                    raise RuntimeError("bn_stats were started on a different tensor than the one being normalised")
                x = sx                        # the channels-last tensor the statistics were started on (same values)
            gain, beta, gb_rows, ssb, count, count_dev, mean, invstd, scale, shift = _bn_forward_stats(x, bn, gain, beta,
                                                                                                      opt.bn_stats)
            opt.bn_stats = None
            flags |= L.ICG_PRE_AFFINE
        res = None
        fflags = flags
        if residual is not None:
            res = _cl(residual)
            if opt.res_up:
                fflags |= L.ICG_RES_UPSAMPLE2X
                assert res.shape == (B, Cout, H // 2, W // 2)
            else:
                assert res.shape == (B, Cout, H, W)
        out = _empty_cl(B, Cout, H, W, dev)
        phase = bool(up and (sn.w_up is not None or sn.rs[0]))
        keep_v = bool(ctx.needs_input_grad[1]) and KEEP_WINOGRAD_V
        saved_v = None                 # (V planes of the forward transform, plane count) for the weight gradient
        if down and sn.rs[0]:
            # conv3x3 + avgpool2 in the 25-plane F(4x4,3x3) domain (25/64 of the 4x4-stride-2 form's MACs)
            nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, Hs, Ws, Cin, Cout)
            ws = _bytes(nb, dev)
            L.call("icg_conv2d_down_wino_fprop", x, sn.w_wino, bias, res, out, B, H, W, Cin, Cout,
                   flags | (L.ICG_WINO_KEEP_V if (keep_v and sn.rs[2]) else 0), ws, nb)
            if keep_v and sn.rs[2]:
                saved_v = (_saved_v(ws, 25, B, Hs, Ws, Cin), 25)
        elif down:
            # conv3x3 + avgpool2 as one 4x4 / stride-2 conv at the pooled resolution (2.25x fewer MACs)
            L.call("icg_conv2d_down_fprop", x, sn.w_down, bias, res, out, B, H, W, Cin, Cout, flags)
        elif phase and sn.rs[0]:
            assert res is None, "the upsample-fused path has no residual epilogue (GBlock conv1 has none)"
            nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cin, Cout)
            ws = _bytes(nb, dev)
            # V planes for the weight gradient: the composite leaves them in the workspace anyway; as a by-product of the fused
            # kernel the 25-plane V of an upsampled input costs more (+1.4 ms at 192 -> 96 @256) than transforming x again in the
            # backward pass (1.0 ms), so it is not kept there
            keep_up = bool(keep_v and sn.rs[2] and not fwino_applies(B, H, W, Cin, Cout))
            L.call("icg_conv2d_up_wino_fprop", x, sn.w_wino, bias, out, scale, shift, ssb, B, Hs, Ws, Cin, Cout,
                   (flags & ~L.ICG_UPSAMPLE2X) | (L.ICG_WINO_KEEP_V if keep_up else 0), ws, nb)
            if keep_up:
                saved_v = (_saved_v(ws, 25, B, H, W, Cin), 25)
        elif phase:
            # nearest-x2 + 3x3 as 4 phases of 2x2 taps on the source tensor (2.25x fewer MACs)
            assert res is None, "the phase path has no residual epilogue (GBlock conv1 has none)"
            L.call("icg_conv2d_up_fprop", x, sn.w_up, bias, out, scale, shift, ssb, B, Hs, Ws, Cin, Cout,
                   flags & ~L.ICG_UPSAMPLE2X)
        elif sn.w_wino is not None and not up:
            # wide 3x3 stride-1 layer: Winograd F(2x2,3x3), 16/36 of the multiply-adds (csrc/winograd.hip)
            v = _wino_fprop(x, sn.w_wino, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, fflags, sn.wino_m,
                            keep_v and winograd_wgrad_tile(Cin, Cout, H, W, B) == 4)
            saved_v = (v, 36) if v is not None else None
        else:
            _conv_fprop(x, sn.w_ohwi, bias, res, out, scale, shift, ssb, B, H, W, Cin, Cout, R, fflags)
        ctx.saved_v = saved_v
        ctx.phase, ctx.down = phase, down
        ctx.opt, ctx.flags, ctx.dims = opt, flags, (B, Cin, Hs, Ws, H, W, Cout, R, gb_rows, ssb, count)
        ctx.has = (bias is not None, residual is not None, gain is not None, beta is not None)
        ctx.weight_like = weight
        ctx.save_for_backward(x, scale, shift, mean, invstd, gain, count_dev)
        if opt.chain:
            # gradient chain: the caller feeds `x_next` (the same values as x) to the next consumer of x instead of x itself, so
            # that the consumers' data gradients arrive here one after the other and are ADDED IN THE EPILOGUE of this layer's
            # data-gradient convolution (residual operand) -- instead of autograd summing separate tensors with elementwise adds
            # (three per attention block and backward pass: theta / phi / g / the residual path all read x)
            return out, x_in
        return out

    @staticmethod
    def backward(ctx, dout, dcarry=None):
        x, scale, shift, mean, invstd, gain, count_dev = ctx.saved_tensors
        opt, flags = ctx.opt, ctx.flags
        if dcarry is not None:
            dcarry = _cl(dcarry)
        sn, bn = opt.sn, opt.bn
        B, Cin, Hs, Ws, H, W, Cout, R, gb_rows, ssb, count = ctx.dims
        has_bias, has_res, has_gain, has_beta = ctx.has
        dev = x.device
        dout = _cl(dout)
        need = ctx.needs_input_grad
        dx = dweight = dbias = dres = dgain = dbeta = None
        bn_state = None
        entry = None
        if WGRAD_SIDE_STREAM and dout.is_cuda and need[0] and need[1]:
            entry = torch.cuda.Event()
            entry.record()               # dout (and everything the forward saved) is ready on the main stream here
        if need[0] or (bn is not None and (need[4] or need[5])):
            # ReLU-only prologue (every D layer): its backward, dx = (x > 0) ? da : 0, runs in the epilogue of the data-gradient
            # convolution (ICG_RES_RELU_MASK) -- da is never written and no separate pass reads x and da
            mask = bn is None and opt.relu and not opt.upsample and FUSE_RELU_BACKWARD
            if ctx.down:
                if (sn.w_wino_dgrad if sn.rs[1] else sn.w_down_dgrad) is None:
                    raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
                da = _empty_cl(B, Cin, Hs, Ws, dev)          # full (input) resolution
                if sn.rs[1]:
                    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, Hs, Ws, Cout, Cin)
                    if mask:
                        L.call("icg_conv2d_down_wino_dgrad_relu", dout, sn.w_wino_dgrad, x, da, B, H, W, Cin, Cout,
                               _bytes(nb, dev), nb)
                    else:
                        L.call("icg_conv2d_down_wino_dgrad", dout, sn.w_wino_dgrad, da, B, H, W, Cin, Cout, _bytes(nb, dev), nb)
                elif mask:
                    L.call("icg_conv2d_down_dgrad_relu", dout, sn.w_down_dgrad, x, da, B, H, W, Cin, Cout)
                else:
                    L.call("icg_conv2d_down_dgrad", dout, sn.w_down_dgrad, da, B, H, W, Cin, Cout)
            elif ctx.phase:
                mask = False
                if (sn.w_wino_dgrad if sn.rs[1] else sn.w_up_dgrad) is None:
                    raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
                da = _empty_cl(B, Cin, Hs, Ws, dev)          # already at source resolution (upsample adjoint folded)
                if sn.rs[1]:
                    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cout, Cin)
                    L.call("icg_conv2d_up_wino_dgrad", dout, sn.w_wino_dgrad, da, B, Hs, Ws, Cin, Cout, _bytes(nb, dev), nb)
                else:
                    L.call("icg_conv2d_up_dgrad", dout, sn.w_up_dgrad, da, B, Hs, Ws, Cin, Cout)
                flags = flags & ~L.ICG_UPSAMPLE2X
            else:
                if sn.w_dgrad is None:
                    raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
                da = _empty_cl(B, Cin, H, W, dev)
                mres, mflag = (x, L.ICG_RES_RELU_MASK) if mask else (None, 0)
                if dcarry is not None and not mask and bn is None and not opt.relu and not opt.upsample:
                    mres, mflag, dcarry = dcarry, 0, None      # the later consumers' gradient of x: residual add in the epilogue
                if sn.w_wino_dgrad is not None:
                    _wino_fprop(dout, sn.w_wino_dgrad, None, mres, da, None, None, 0, B, H, W, Cout, Cin, mflag, sn.wino_m)
                else:
                    _conv_fprop(dout, sn.w_dgrad, None, mres, da, None, None, 0, B, H, W, Cout, Cin, R, mflag)
            if bn is not None:
                # stages 1-2 now (+ the cross-replica all-reduce of the channel sums, asynchronous); stages 3-4 after the
                # weight / bias gradients below, which do not depend on them and overlap the collective
                bn_state = _bn_backward_begin(x, da, bn, gain, scale, shift, ssb, mean, invstd, gb_rows, count_dev, flags,
                                              (B, Cin, Hs, Ws))
                bn_flags = flags
            elif mask:
                dx = da
            elif opt.relu or (opt.upsample and not ctx.phase):
                dx = _empty_cl(B, Cin, Hs, Ws, dev)
                L.call("icg_bn_bwd_apply", x, da, None, None, 0, None, None, None, B, Hs, Ws, Cin, flags, dx)
            else:
                dx = da
        def weight_and_bias_gradients():
            """dweight (through the spectral-norm backward) and dbias of this layer: everything that depends only on the saved
            forward tensors and dout, not on the data gradient."""
            dweight = dbias = None
            if need[1] and (ctx.down or ctx.phase) and sn.rs[2]:
                # weight gradient of the resample-fused layer in the 25-plane domain: plain HWIO 3x3 result
                dw_hwio = _f32(9 * Cin * Cout, dev)
                Hf, Wf = (Hs, Ws) if ctx.down else (H, W)                # full resolution of the layer
                if ctx.saved_v is not None and ctx.saved_v[1] == 25:
                    dbias = _wgrad_from_v(ctx.saved_v[0], dout, dw_hwio, B, Hf, Wf, Cin, Cout, 25, 1 if ctx.down else 0,
                                          0.25 if ctx.down else 1.0, has_bias and need[2])
                    ctx.saved_v = None
                elif ctx.down:
                    nb = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, Hs, Ws, Cin, Cout)
                    L.call("icg_conv2d_down_wino_wgrad", x, dout, dw_hwio, B, H, W, Cin, Cout, ctx.flags, _bytes(nb, dev), nb)
                else:
                    nb = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, H, W, Cin, Cout)
                    L.call("icg_conv2d_up_wino_wgrad", x, dout, dw_hwio, scale, shift, ssb, B, Hs, Ws, Cin, Cout,
                           ctx.flags & ~L.ICG_UPSAMPLE2X, _bytes(nb, dev), nb)
                dweight = _sn_backward(dw_hwio, None, sn, ctx.weight_like)
            elif need[1] and ctx.down:
                nb = L.query("icg_conv2d_down_wgrad_workspace_bytes", B, H, W, Cin, Cout)
                ws = _bytes(nb, dev)
                dw_down = _f32(16 * Cin * Cout, dev)
                L.call("icg_conv2d_down_wgrad", x, dout, dw_down, B, H, W, Cin, Cout, ctx.flags, ws, nb)
                dweight = _sn_backward(None, None, sn, ctx.weight_like, dw_down=dw_down)
            elif need[1] and ctx.phase:
                nb = L.query("icg_conv2d_up_wgrad_workspace_bytes", B, Hs, Ws, Cin, Cout)
                ws = _bytes(nb, dev)
                dw_up = _f32(16 * Cin * Cout, dev)
                L.call("icg_conv2d_up_wgrad", x, dout, dw_up, scale, shift, ssb, B, Hs, Ws, Cin, Cout,
                       ctx.flags & ~L.ICG_UPSAMPLE2X, ws, nb)
                dweight = _sn_backward(None, None, sn, ctx.weight_like, dw_up=dw_up)
            elif need[1]:
                dw_hwio = _f32(R * R * Cin * Cout, dev)
                wt = winograd_wgrad_tile(Cin, Cout, H, W, B) if sn.w_wino is not None else 0
                if wt == 4 and ctx.saved_v is not None and ctx.saved_v[1] == 36:
                    dbias = _wgrad_from_v(ctx.saved_v[0], dout, dw_hwio, B, H, W, Cin, Cout, 36, 0, 1.0, has_bias and need[2])
                    ctx.saved_v = None
                elif wt:
                    # wide 3x3 stride-1 layer: weight gradient through the Winograd domain (16/36 or 9/36 of the MACs)
                    v = "wino4" if wt == 4 else "wino"
                    nb = L.query("icg_conv2d_%s_wgrad_workspace_bytes" % v, B, H, W, Cin, Cout)
                    L.call("icg_conv2d_%s_wgrad" % v, x, dout, dw_hwio, scale, shift, ssb, B, H, W, Cin, Cout, ctx.flags,
                           _bytes(nb, dev), nb)
                else:
                    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, R)
                    ws = _bytes(nb, dev)
                    L.call("icg_conv2d_wgrad", x, dout, dw_hwio, scale, shift, ssb, B, H, W, Cin, Cout, R, ctx.flags, ws, nb)
                dweight = _sn_backward(dw_hwio, None, sn, ctx.weight_like)
            if has_bias and need[2] and dbias is None:
                rows = B * H * W
                nb = L.query("icg_colsum_workspace_bytes", rows, Cout)
                ws = _bytes(nb, dev)
                dbias = _f32(Cout, dev)
                L.call("icg_colsum", dout, rows, Cout, dbias, ws, nb)
            return dweight, dbias

        if need[1] or (has_bias and need[2]):
            side = _wgrad_stream(dev) if (need[1] and need[0] and entry is not None) else None
            if side is None:
                dweight, dbias = weight_and_bias_gradients()
            else:
                # weight gradient on the side stream, concurrently with the data-gradient kernels queued on the main stream above
                # (see WGRAD_SIDE_STREAM).  Cross-stream tensor lifetimes are declared to the caching allocator.
                main = torch.cuda.current_stream(dev)
                held = [x, dout, scale, shift, ctx.saved_v[0] if ctx.saved_v is not None else None,
                        sn.w_ohwi, sn.u, sn.v, sn.sigma]
                side.wait_event(entry)
                with torch.cuda.stream(side):
                    dweight, dbias = weight_and_bias_gradients()
                    done = torch.cuda.Event()
                    done.record(side)
                for t in held:
                    if t is not None:
                        t.record_stream(side)
                main.wait_event(done)
                for t in (dweight, dbias):
                    if t is not None:
                        t.record_stream(main)
        if has_res and need[3]:
            if opt.res_up:
                dres = _empty_cl(B, Cout, H // 2, W // 2, dev)
                L.call("icg_sumpool2_fwd", dout, dres, B, H, W, Cout)
            else:
                dres = dout
        if bn_state is not None:
            dx, dgain, dbeta = _bn_backward_end(bn_state, x, da, bn, scale, shift, ssb, mean, invstd, gb_rows, count,
                                                bn_flags, has_gain, has_beta, (B, Cin, Hs, Ws))
        if dcarry is not None and dx is not None:
            dx = dx + dcarry                   # (layer forms whose epilogue is taken: fall back to the elementwise add)
        elif dcarry is not None and need[0]:
            dx = dcarry
        if not need[0]:
            dx = None
        return dx, dweight, dbias, dres, dgain, dbeta, None


def bn_stats_begin(x, bn: BNOpt) -> Optional[BNStats]:
    """Start the training-mode statistics of `x`.  With cross-replica BN (sync_bn) the packed payload [sum x | sum x^2 | n]
    (icg_bn_sync_pack: common origin, count on the device) is all-reduced asynchronously: RCCL runs it on its own stream
    behind an event on the current one, so every kernel launched between this call and the consumer's BNStats.wait()
    overlaps the (latency-bound, <= 12 KB) collective.  Replaces the master/slave exchange of
    sync_batchnorm/batchnorm.py:148-193.  Returns None in eval mode (running statistics)."""
    if not bn.training:
        return None
    x = _cl(x)
    B, C, Hs, Ws = x.shape
    dev = x.device
    rows = B * Hs * Ws
    nb = L.query("icg_bn_workspace_bytes", rows, C)
    ws = _bytes(nb, dev)
    L.call("icg_bn_partial_stats", x, bn.running_mean, rows, C, ws, nb)
    sums = torch.empty(2 * C, device=dev, dtype=torch.float64)
    L.call("icg_bn_reduce_partials", ws, rows, C, sums)
    if not _sync_enabled(bn):
        return BNStats(x, sums, bn.running_mean, float(rows))
    payload = torch.empty(2 * C + 1, device=dev, dtype=torch.float64)
    L.call("icg_bn_sync_pack", sums, bn.running_mean, float(rows), C, payload)
    work = dist.all_reduce(payload, group=_group(bn), async_op=True)
    return BNStats(x, payload, None, 0.0, work)


def _bn_forward_stats(x, bn: BNOpt, gain, beta, stats: Optional[BNStats] = None):
    """Statistics + finalize shared by the fused and the stand-alone normalisation.
    -> (..., count, count_dev, ...): count is the host-side element count (0.0 under cross-replica BN, where the global count
    stays on the device as the 1-element tensor count_dev)."""
    B, C, Hs, Ws = x.shape
    dev = x.device
    gb_rows = 1
    if gain is not None:
        gain = gain.contiguous()
        gb_rows = gain.shape[0] if gain.dim() == 2 else 1
    if beta is not None:
        beta = beta.contiguous()
    mean, invstd = _f32(C, dev), _f32(C, dev)
    scale, shift = _f32(gb_rows * C, dev), _f32(gb_rows * C, dev)
    ssb = C if gb_rows > 1 else 0
    count = float(B * Hs * Ws)
    sums = shift_k = count_dev = None
    if bn.training and stats is None and not _sync_enabled(bn):
        # single replica: partial sums, then their reduction and the finalize step as ONE launch (icg_bn_reduce_finalize)
        x = _cl(x)
        rows = B * Hs * Ws
        nb = L.query("icg_bn_workspace_bytes", rows, C)
        ws = _bytes(nb, dev)
        L.call("icg_bn_partial_stats", x, bn.running_mean, rows, C, ws, nb)
        L.call("icg_bn_reduce_finalize", ws, rows, C, bn.running_mean, bn.running_mean, bn.running_var, float(bn.momentum),
               float(bn.eps), gain, beta, gb_rows, float(bn.gain_offset), mean, invstd, scale, shift)
        return gain, beta, gb_rows, ssb, count, count_dev, mean, invstd, scale, shift
    if bn.training:
        if stats is None:
            stats = bn_stats_begin(x, bn)
        elif stats.x.data_ptr() != x.data_ptr() or stats.x.shape != x.shape:
            raise RuntimeError("bn_stats were started on a different tensor than the one being normalised")
        stats.wait()
        sums, shift_k, count = stats.sums, stats.shift, stats.count
        if count <= 0.0:
            count_dev = sums[2 * C:]
    else:
        shift_k = bn.running_mean
    L.call("icg_bn_finalize", sums, shift_k, count, bn.running_mean, bn.running_var, float(bn.momentum),
           float(bn.eps), int(bn.training), gain, beta, gb_rows, float(bn.gain_offset), C, mean, invstd, scale, shift)
    return gain, beta, gb_rows, ssb, count, count_dev, mean, invstd, scale, shift


def _bn_backward_begin(x, da, bn: BNOpt, gain, scale, shift, ssb, mean, invstd, gb_rows, count_dev, flags, dims):
    """Stages 1-2 of the BN backward (include/icgan_hip.h) and, under cross-replica BN, the start of the asynchronous
    all-reduce of the per-channel sums; -> state for _bn_backward_end.  Work launched in between overlaps the collective."""
    B, C, Hs, Ws = dims
    dev = x.device
    nb = L.query("icg_bn_bwd_workspace_bytes", B, Hs, Ws, C)
    ws = _bytes(nb, dev)
    sd, sx = _f32(B * C, dev), _f32(B * C, dev)
    L.call("icg_bn_bwd_reduce", x, da, scale, shift, ssb, mean, B, Hs, Ws, C, flags, ws, nb, sd, sx)
    chan = work = None
    if bn.training:
        chan = torch.empty(2 * C + 1, device=dev, dtype=torch.float64)
        # (the kernel owns chan[0 .. 2 C); slot 2 C is the element count of the cross-replica convention, written below)
        L.call("icg_bn_bwd_channel_sums", sd, sx, gain, gb_rows, float(bn.gain_offset), invstd, B, C, chan[: 2 * C])
        if count_dev is not None:
            chan[2 * C:].copy_(count_dev)            # global element count of the forward (device side)
            if _sync_enabled(bn):
                work = dist.all_reduce(chan[: 2 * C], group=_group(bn), async_op=True)
    return sd, sx, chan, work


def _bn_backward_end(state, x, da, bn: BNOpt, scale, shift, ssb, mean, invstd, gb_rows, count, flags, has_gain, has_beta,
                     dims):
    """Stages 3-4: -> dx, dgain, dbeta."""
    sd, sx, chan, work = state
    B, C, Hs, Ws = dims
    dev = x.device
    if work is not None:
        work.wait()
    dgain = _f32(gb_rows * C, dev).view(gb_rows, C) if has_gain else None
    dbeta = _f32(gb_rows * C, dev).view(gb_rows, C) if has_beta else None
    coef_a, coef_b = _f32(C, dev), _f32(C, dev)
    L.call("icg_bn_bwd_coefs", sd, sx, chan, invstd, float(count), int(bn.training), gb_rows, B, C, dgain, dbeta,
           coef_a, coef_b)
    dx = _empty_cl(B, C, Hs, Ws, dev)
    L.call("icg_bn_bwd_apply", x, da, scale, shift, ssb, mean, coef_a, coef_b, B, Hs, Ws, C, flags, dx)
    return dx, dgain, dbeta


def _bn_backward(x, da, bn: BNOpt, gain, scale, shift, ssb, mean, invstd, gb_rows, count, count_dev, flags, has_gain,
                 has_beta, dims):
    """Shared BN backward: returns dx, dgain, dbeta (see include/icgan_hip.h, stages 1-4)."""
    st = _bn_backward_begin(x, da, bn, gain, scale, shift, ssb, mean, invstd, gb_rows, count_dev, flags, dims)
    return _bn_backward_end(st, x, da, bn, scale, shift, ssb, mean, invstd, gb_rows, count, flags, has_gain, has_beta, dims)


class NormActFn(Function):
    """Stand-alone  y = relu?(BN(x)*gain + bias)  (ccbn.forward / bn.forward called outside a fused block)."""

    @staticmethod
    def forward(ctx, x, gain, beta, bn: BNOpt, relu: bool):
        _require_gpu(x)
        x = _cl(x)
        B, C, H, W = x.shape
        gain, beta, gb_rows, ssb, count, count_dev, mean, invstd, scale, shift = _bn_forward_stats(x, bn, gain, beta)
        flags = L.ICG_PRE_AFFINE | (L.ICG_PRE_RELU if relu else 0)
        y = torch.empty_like(x)
        L.call("icg_bn_apply", x, scale, shift, ssb, B, H * W, C, flags, y)
        ctx.bn, ctx.meta = bn, (gb_rows, ssb, count, flags, gain is not None, beta is not None, (B, C, H, W))
        ctx.save_for_backward(x, scale, shift, mean, invstd, gain, count_dev)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mean, invstd, gain, count_dev = ctx.saved_tensors
        gb_rows, ssb, count, flags, has_gain, has_beta, dims = ctx.meta
        dx, dgain, dbeta = _bn_backward(x, _cl(dy), ctx.bn, gain, scale, shift, ssb, mean, invstd, gb_rows, count,
                                        count_dev, flags, has_gain, has_beta, dims)
        return dx, dgain, dbeta, None, None


def norm_act(x, bn: BNOpt, gain, beta, relu=False):
    g2 = gain if gain is None or gain.dim() == 2 else gain.view(1, -1)
    b2 = beta if beta is None or beta.dim() == 2 else beta.view(1, -1)
    return NormActFn.apply(x, g2, b2, bn, relu)


def fused_conv(x, weight, bias, sn: SNState, *, relu=False, upsample=False, residual=None, res_up=False,
               bn: Optional[BNOpt] = None, gain=None, beta=None, downsample=False, bn_stats: Optional[BNStats] = None,
               chain=False):
    """conv(act(x)) with act = [BN affine] -> [ReLU] -> [nearest x2]; `gain`/`beta` are [B,C] (ccbn) or [C] (bn);
    downsample=True appends the 2x2 average pool (residual is then at the pooled resolution)."""
    opt = ConvOpt(sn=sn, relu=relu, upsample=upsample, res_up=res_up, bn=bn, downsample=downsample, bn_stats=bn_stats,
                  chain=bool(chain))
    if sn.handle is not None:
        weight = sn.handle                  # grouped spectral-norm backward: this node returns the raw gradient of W / sigma
    if bn is not None:
        g2 = gain if gain is None or gain.dim() == 2 else gain.view(1, -1)
        b2 = beta if beta is None or beta.dim() == 2 else beta.view(1, -1)
        out = FusedConvFn.apply(x, weight, bias, residual, g2, b2, opt)
        return out
    return FusedConvFn.apply(x, weight, bias, residual, None, None, opt)


def linear(x2d: torch.Tensor, weight, bias, sn: SNState):
    """F.linear(x, W/sigma, b) as a 1x1 'convolution' over M rows (reference layers.py:164-165)."""
    m, k = x2d.shape
    out = fused_conv(x2d.contiguous().view(m, k, 1, 1), weight, bias, sn)
    return out.reshape(m, sn.rows)


# ----------------------------------------------------------------------------------------------
# the conditional-BN projections of a block as one launch per direction  (reference layers.py:367-374)
# ----------------------------------------------------------------------------------------------
GROUPED_CCBN = os.environ.get("ICG_CCBN_GROUP", "1") != "0"


def _linear_group(mode, M, K, items):
    """icg_linear_group: items = [(x, w, dy, out, N), ...] (tensors or None)."""
    import ctypes
    arr = (L.LinearItem * len(items))()
    for d, (x, w, dy, out, n) in zip(arr, items):
        d.x, d.w, d.dy, d.out = (t.data_ptr() if t is not None else None for t in (x, w, dy, out))
        d.N = int(n)
    L.call("icg_linear_group", ctypes.cast(arr, ctypes.c_void_p), len(items), int(M), int(K), int(mode))


class CcbnAffineFn(Function):
    """gain(y), bias(y) of bn1 and gain(y), bias(y) of bn2 of a GBlock -- four spectrally normalised linear layers without bias
    applied to the same conditioning vector y [B, K] (reference layers.py:367-374, called from 542-552) -- as ONE launch forward, one
    for the four weight gradients and one for the gradient of y (summed inside the kernel) instead of four launches each way plus
    three elementwise adds.  Inputs: y, the SNStates, the weights (or their SNGroupFn handles)."""

    @staticmethod
    def forward(ctx, y, sns, *ws):
        y = y.contiguous().float()
        M, K = y.shape
        _require_gpu(y)
        outs = [torch.empty(M, sn.rows, device=y.device, dtype=torch.float32) for sn in sns]
        _linear_group(0, M, K, [(y, sn.w_ohwi, None, o, sn.rows) for sn, o in zip(sns, outs)])
        ctx.sns, ctx.likes = sns, ws
        ctx.save_for_backward(y)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        (y,) = ctx.saved_tensors
        M, K = y.shape
        sns, need = ctx.sns, ctx.needs_input_grad
        douts = [d.contiguous().float() for d in douts]
        dy = None
        if need[0]:
            if any(sn.w_dgrad is None for sn in sns):
                raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
            dy = torch.empty_like(y)
            _linear_group(2, M, K, [(None, sn.w_dgrad, d, dy if i == 0 else None, sn.rows) for i, (sn, d) in enumerate(zip(sns, douts))])
        dws = [None] * len(sns)
        todo = [i for i in range(len(sns)) if need[2 + i]]
        if todo:
            raws = {i: _f32(K * sns[i].rows, y.device) for i in todo}
            _linear_group(1, M, K, [(y, None, douts[i], raws[i], sns[i].rows) for i in todo])
            for i in todo:
                dws[i] = _sn_backward(raws[i], None, sns[i], ctx.likes[i])
        return (dy, None) + tuple(dws)


# ----------------------------------------------------------------------------------------------
# SN embedding  (reference layers.py:171-200)
# ----------------------------------------------------------------------------------------------
class SNEmbeddingFn(Function):
    @staticmethod
    def forward(ctx, idx, weight, sn: SNState):
        w = sn.w_ohwi.view(sn.rows, sn.cin)
        ctx.sn, ctx.weight_like = sn, weight
        ctx.save_for_backward(idx)
        return w.index_select(0, idx.reshape(-1)).view(*idx.shape, sn.cin)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        sn = ctx.sn
        # dW[r] = sum over the samples with label r of dout[sample], IN SAMPLE ORDER: one-hot^T [rows x n] times dout [n x cin] on the HIP
        # GEMM (a fixed-order K loop).  Round 6: this was `index_add_`, whose atomic adds land in whatever order the hardware schedules
        # them -- with repeated labels in a batch (128 draws from 1000 classes: almost always) the gradient of D.embed differed in the last
        # bits from process to process, the one nondeterministic operation of the step (profiles/r06_step_determinism.txt).  The
        # reference's nn.Embedding backward sorts the indices and is deterministic as well (layers.py:171-200 -> F.embedding).
        flat = idx.reshape(-1)
        n = int(flat.numel())
        d2 = dout.reshape(n, sn.cin).float().contiguous()
        dw_ = torch.empty(sn.rows, sn.cin, device=dout.device, dtype=torch.float32)
        if n * sn.rows <= (1 << 26):
            onehot = torch.zeros(n, sn.rows, device=dout.device, dtype=torch.float32)
            onehot.scatter_(1, flat.view(n, 1), 1.0)                  # (one element per row: no two writes meet)
            L.call("icg_gemm_batched", onehot, d2, dw_, sn.rows, sn.cin, n, 1, 0, 0, 0, 0, 1, 1.0)
        else:     # a table too large for the one-hot form: ATen's sort-based (deterministic) embedding backward
            dw_ = torch.ops.aten.embedding_dense_backward(d2, flat, sn.rows, -1, False)
        return None, _sn_backward(None, dw_, sn, ctx.weight_like), None


# ----------------------------------------------------------------------------------------------
# pooling / pointwise
# ----------------------------------------------------------------------------------------------
class AvgPool2Fn(Function):
    """y = avgpool2x2(x) (+ add)   (nn.AvgPool2d(2), BigGAN.py:528; residual add of layers.py:613).
    chain=True: also returns x as a second output (see FusedConvFn: gradient chain) -- the next consumer's gradient of x is added
    inside the pooling backward kernel (icg_avgpool2_bwd_add)."""

    @staticmethod
    def forward(ctx, x, add, chain=False):
        x_in = x
        x = _cl(x)
        B, C, H, W = x.shape
        y = _empty_cl(B, C, H // 2, W // 2, x.device)
        a = _cl(add) if add is not None else None
        L.call("icg_avgpool2_fwd", x, a, y, B, H, W, C)
        ctx.dims = (B, C, H, W)
        ctx.has_add = add is not None
        if chain:
            return y, (x_in if x.data_ptr() == x_in.data_ptr() and x.dtype == x_in.dtype else x)
        return y

    @staticmethod
    def backward(ctx, dy, dcarry=None):
        B, C, H, W = ctx.dims
        dy = _cl(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty_cl(B, C, H, W, dy.device)
            if dcarry is not None:
                L.call("icg_avgpool2_bwd_add", dy, _cl(dcarry), dx, B, H, W, C)
            else:
                L.call("icg_avgpool2_bwd", dy, dx, B, H, W, C)
        return dx, (dy if ctx.has_add and ctx.needs_input_grad[1] else None), None


class MaxPool2Fn(Function):
    """F.max_pool2d(x, [2,2])  (layers.py:230-231)."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        B, C, H, W = x.shape
        y = _empty_cl(B, C, H // 2, W // 2, x.device)
        L.call("icg_maxpool2_fwd", x, y, B, H, W, C)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dx = _empty_cl(B, C, H, W, x.device)
        L.call("icg_maxpool2_bwd", x, _cl(dy), dx, B, H, W, C)
        return dx


class TanhFn(Function):
    """torch.tanh tail of the generator (BigGAN.py:386)."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        y = torch.empty_like(x)
        L.call("icg_tanh_fwd", x, y, x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _cl(dy)
        dx = torch.empty_like(y)
        L.call("icg_tanh_bwd", y, dy, dx, y.numel())
        return dx


class ReluSumPoolFn(Function):
    """h = sum(relu(x), [2,3])   (BigGAN.py:625)."""

    @staticmethod
    def forward(ctx, x):
        x = _cl(x)
        B, C, H, W = x.shape
        y = torch.empty(B, C, device=x.device, dtype=torch.float32)
        L.call("icg_relu_sumpool_fwd", x, y, B, H * W, C)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        L.call("icg_relu_sumpool_bwd", x, dy.contiguous(), dx, B, H * W, C)
        return dx


class ScaleAddFn(Function):
    """gamma * o + x   (layers.py:244)."""

    @staticmethod
    def forward(ctx, gamma, o, x):
        o, x = _cl(o), _cl(x)
        out = torch.empty_like(x)
        g = gamma.detach().reshape(1).contiguous()
        L.call("icg_scale_add_fwd", g, o, x, out, x.numel())
        ctx.save_for_backward(g, o)
        ctx.gshape = gamma.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        g, o = ctx.saved_tensors
        dout = _cl(dout)
        d_o = torch.empty_like(o)
        dgamma = _f32(1, o.device)
        nb = 2048 * 8
        L.call("icg_scale_add_bwd", g, o, dout, d_o, dgamma, o.numel(), _bytes(nb, o.device), nb)
        return dgamma.view(ctx.gshape), d_o, dout


# ----------------------------------------------------------------------------------------------
# attention core: beta = softmax(theta^T phi); o = g beta^T   (layers.py:233-243)
# ----------------------------------------------------------------------------------------------
FUSED_ATTENTION_SCORES = True       # False: scores GEMM + stand-alone softmax (measurement switch, bench.py --no-fused-attention)


class AttnCoreFn(Function):
    @staticmethod
    def forward(ctx, theta, phi, g):
        theta, phi, g = _cl(theta), _cl(phi), _cl(g)
        B, d, H, W = theta.shape
        dv = g.shape[1]
        n, m = H * W, phi.shape[2] * phi.shape[3]
        dev = theta.device
        # NHWC memory: theta = Q [B][n][d], phi = K [B][m][d], g = V [B][m][dv]
        beta = torch.empty(B, n, m, device=dev, dtype=torch.float32)
        if FUSED_ATTENTION_SCORES and L.query("icg_attn_scores_softmax_applies", n, m, d):
            # scores + softmax in one kernel (csrc/attn.hip): the [B][n][m] score tensor is never written
            L.call("icg_attn_scores_softmax", theta, phi, beta, B, n, m, d)
        else:
            s = torch.empty_like(beta)
            L.call("icg_gemm_batched", theta, phi, s, n, m, d, 0, 1, n * d, m * d, n * m, B, 1.0)
            L.call("icg_softmax_fwd", s, beta, B * n, m)
            del s
        o = _empty_cl(B, dv, H, W, dev)
        L.call("icg_gemm_batched", beta, g, o, n, dv, m, 0, 0, n * m, m * dv, n * dv, B, 1.0)
        ctx.save_for_backward(theta, phi, g, beta)
        return o

    @staticmethod
    def backward(ctx, do):
        theta, phi, g, beta = ctx.saved_tensors
        B, d, H, W = theta.shape
        dv = g.shape[1]
        n, m = H * W, phi.shape[2] * phi.shape[3]
        dev = theta.device
        do = _cl(do)
        dg = torch.empty_like(g)           # dV = beta^T dO
        if ATTN_DV_PLANE_GEMM and dv % 96 == 0 and n % 32 == 0 and m % 4 == 0:
            # a weight-gradient-shaped contraction (K = n = 4096 query rows per image, [m x dv] = [1024 x 192 | 96] outputs): the second-
            # generation TN plane GEMM (csrc/pgemm.hip, LDS-DMA operands, K split in two slabs summed in fixed order) with one "plane" per
            # image.  On the first-generation kernel this GEMM ran at ~53 TFLOP/s -- 4 launches x 1.93 ms per cfg3 step
            # (`icg_gemm_kernel<1,1,3,2>` in profiles/r05_bench_cfg3_kernel_stats.csv).
            nbw = L.query("icg_plane_gemm_tn_workspace_bytes", m, dv, n, B)
            L.call("icg_plane_gemm_tn", beta, do, dg, m, dv, n, B, _bytes(nbw, dev), nbw)
        else:
            L.call("icg_gemm_batched", beta, do, dg, m, dv, n, 1, 0, n * m, n * dv, m * dv, B, 1.0)
        ds = torch.empty_like(beta)
        if FUSED_ATTENTION_SCORES and L.query("icg_attn_dscores_applies", n, m, dv):
            # dP = dO V^T with the softmax backward in its epilogue (csrc/attn.hip): the [B][n][m] dP tensor is never written
            L.call("icg_attn_dscores", do, g, beta, ds, B, n, m, dv)
        else:
            dbeta = torch.empty_like(beta)     # dP = dO V^T
            L.call("icg_gemm_batched", do, g, dbeta, n, m, dv, 0, 1, n * dv, m * dv, n * m, B, 1.0)
            L.call("icg_softmax_bwd", beta, dbeta, ds, B * n, m)
            del dbeta
        dtheta = torch.empty_like(theta)   # dQ = dS K
        L.call("icg_gemm_batched", ds, phi, dtheta, n, d, m, 0, 0, n * m, m * d, n * d, B, 1.0)
        dphi = torch.empty_like(phi)       # dK = dS^T Q
        L.call("icg_gemm_batched", ds, theta, dphi, m, d, n, 1, 0, n * m, n * d, m * d, B, 1.0)
        return dtheta, dphi, dg


FUSED_ATTENTION_PROJECTIONS = os.environ.get("ICG_ATTN_PROJ", "1") != "0"
ATTN_DV_PLANE_GEMM = os.environ.get("ICG_ATTN_DV_TN", "1") != "0"      # dV = beta^T dO on the TN plane GEMM (measurement switch)


def attn_projections_apply(x, d, dv) -> bool:
    """The stacked form of theta / phi / g (AttnProjFn): vector widths of icg_attn_split_pool and an even resolution."""
    return bool(FUSED_ATTENTION_PROJECTIONS and x.dim() == 4 and d % 4 == 0 and dv % 4 == 0 and x.shape[2] % 2 == 0
                and x.shape[3] % 2 == 0)


class AttnProjFn(Function):
    """theta, max-pooled phi and max-pooled g of the attention block (reference layers.py:217-231) from ONE 1x1 convolution whose
    weight stacks the three spectrally normalised matrices: one GEMM with 2 d + dv columns instead of three with d / d / dv (the
    narrow ones fill a quarter or half of an MFMA tile), x read once; backward: one data-gradient GEMM (the later consumers'
    gradient of x added in its epilogue, as FusedConvFn's chain does) and one weight-gradient GEMM whose row blocks are the three
    layers' gradients.  Returns (theta [B,d,H,W], phi [B,d,H/2,W/2], g [B,dv,H/2,W/2], x handed on to the residual path)."""

    @staticmethod
    def forward(ctx, x, w_theta, w_phi, w_g, sns):
        x_in = x
        x = _cl(x)
        B, C, H, W = x.shape
        d, dv = sns[0].rows, sns[2].rows
        assert sns[1].rows == d and all(s.cin == C and s.R == 1 for s in sns)
        Ct = 2 * d + dv
        dev = x.device
        w_cat = torch.cat([s.w_ohwi.view(s.rows, C) for s in sns], 0)
        y = _empty_cl(B, Ct, H, W, dev)
        _conv_fprop(x, w_cat, None, None, y, None, None, 0, B, H, W, C, Ct, 1, 0)
        theta, phi, g = _empty_cl(B, d, H, W, dev), _empty_cl(B, d, H // 2, W // 2, dev), _empty_cl(B, dv, H // 2, W // 2, dev)
        L.call("icg_attn_split_pool", y, theta, phi, g, B, H, W, d, dv)
        ctx.sns, ctx.dims, ctx.likes = sns, (B, C, H, W, d, dv), (w_theta, w_phi, w_g)
        ctx.save_for_backward(x, y)
        return theta, phi, g, (x_in if x.data_ptr() == x_in.data_ptr() and x.dtype == x_in.dtype else x)

    @staticmethod
    def backward(ctx, dtheta, dphi, dg, dcarry=None):
        x, y = ctx.saved_tensors
        B, C, H, W, d, dv = ctx.dims
        Ct = 2 * d + dv
        sns, dev = ctx.sns, x.device
        need = ctx.needs_input_grad
        zeros = lambda c, h, w: torch.zeros(B, c, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        dtheta = _cl(dtheta) if dtheta is not None else zeros(d, H, W)
        dphi = _cl(dphi) if dphi is not None else zeros(d, H // 2, W // 2)
        dg = _cl(dg) if dg is not None else zeros(dv, H // 2, W // 2)
        dy = _empty_cl(B, Ct, H, W, dev)
        L.call("icg_attn_split_pool_bwd", y, dtheta, dphi, dg, dy, B, H, W, d, dv)
        dx = None
        if need[0]:
            if any(s.w_dgrad is None for s in sns):
                raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
            wd_cat = torch.cat([s.w_dgrad.view(C, s.rows) for s in sns], 1)          # [C][2 d + dv]
            dx = _empty_cl(B, C, H, W, dev)
            _conv_fprop(dy, wd_cat, None, _cl(dcarry) if dcarry is not None else None, dx, None, None, 0, B, H, W, Ct, C, 1, 0)
        elif dcarry is not None:
            dx = dcarry
        dws = [None, None, None]
        if any(need[1:4]):
            # the 1x1 weight gradient is symmetric in its operands: with dy as the "input" and x as the "output gradient" the HWIO
            # result [2 d + dv][C] is the stacked gradient in the layers' own OHWI order, its row blocks contiguous
            nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Ct, C, 1)
            dw = _f32(Ct * C, dev)
            L.call("icg_conv2d_wgrad", dy, x, dw, None, None, 0, B, H, W, Ct, C, 1, 0, _bytes(nb, dev), nb)
            r0 = 0
            for i, s in enumerate(sns):
                if need[1 + i]:
                    dws[i] = _sn_backward(None, dw[r0 * C:(r0 + s.rows) * C], s, ctx.likes[i])
                r0 += s.rows
        return dx, dws[0], dws[1], dws[2], None


FUSED_ATTENTION_OUTPUT = os.environ.get("ICG_ATTN_OUT", "1") != "0"


class AttnOutFn(Function):
    """gamma * o(a) + x of the attention block (reference layers.py:242-244) as ONE 1x1 convolution: the weight is gamma * W / sigma
    (scaled on the device, icg_attn_gamma_scale) and x is the residual operand of the epilogue -- o(a) is never written and the
    gamma * o + x pass and its backward (two reads + a write of [B, C, H, W] each) are gone.  Backward: d(a) is the data gradient with
    the scaled weight, the weight gradient dWs of the scaled weight gives dgamma = <dWs, W / sigma> and d(W / sigma) = gamma * dWs
    (icg_attn_gamma_bwd), and x receives dout itself."""

    @staticmethod
    def forward(ctx, a, x, weight, gamma, sn: SNState):
        a, x = _cl(a), _cl(x)
        B, dv, H, W = a.shape
        C = sn.rows
        assert sn.cin == dv and sn.R == 1 and x.shape == (B, C, H, W)
        dev = a.device
        g = gamma.detach().reshape(1).contiguous()
        need_d = sn.w_dgrad is not None
        ws, wds = _f32(C * dv, dev), (_f32(C * dv, dev) if need_d else None)
        L.call("icg_attn_gamma_scale", g, sn.w_ohwi, ws, sn.w_dgrad, wds, C * dv)
        out = _empty_cl(B, C, H, W, dev)
        _conv_fprop(a, ws, None, x, out, None, None, 0, B, H, W, dv, C, 1, 0)
        ctx.sn, ctx.dims, ctx.like, ctx.gshape = sn, (B, dv, H, W, C), weight, gamma.shape
        ctx.save_for_backward(a, g, wds)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, g, wds = ctx.saved_tensors
        B, dv, H, W, C = ctx.dims
        sn, dev = ctx.sn, a.device
        need = ctx.needs_input_grad
        dout = _cl(dout)
        da = dweight = dgamma = None
        if need[0]:
            if wds is None:
                raise RuntimeError("data gradient requested but the layer was prepared without the dgrad layout")
            da = _empty_cl(B, dv, H, W, dev)
            _conv_fprop(dout, wds, None, None, da, None, None, 0, B, H, W, C, dv, 1, 0)
        if need[2] or need[3]:
            if sn.w_dgrad is None:
                raise RuntimeError("weight gradient of the folded projection needs the [Cin][Cout] layout of W / sigma")
            nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, dv, C, 1)
            dws = _f32(dv * C, dev)                                                    # HWIO = [Cin][Cout], the layout of w_dgrad at R = 1
            L.call("icg_conv2d_wgrad", a, dout, dws, None, None, 0, B, H, W, dv, C, 1, 0, _bytes(nb, dev), nb)
            dw_hwio, dgamma = _f32(dv * C, dev), _f32(1, dev)
            L.call("icg_attn_gamma_bwd", g, dws, sn.w_dgrad, dw_hwio, dgamma, dv * C)
            if need[2]:
                dweight = _sn_backward(dw_hwio, None, sn, ctx.like)
            dgamma = dgamma.view(ctx.gshape) if need[3] else None
        return da, (dout if need[1] else None), dweight, dgamma, None


def gemm(a, b, c, m, n, k, trans_a: bool, trans_b: bool, alpha=1.0):
    """Single fp32 GEMM on the MFMA kernel (ortho regularisation, utils.py:1073-1083)."""
    L.call("icg_gemm_batched", a, b, c, m, n, k, int(trans_a), int(trans_b), 0, 0, 0, 1, float(alpha))


# ----------------------------------------------------------------------------------------------
# optimiser / EMA kernels
# ----------------------------------------------------------------------------------------------
def bump_version(*tensors):
    """The HIP kernels write through raw pointers, which autograd's version counters do not see; anything that caches values
    derived from a tensor (the eval-mode W/sigma cache of layers.SN) relies on the counter, so writers bump it explicitly."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def adam_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, step):
    import ctypes
    n = len(params)
    if n == 0:
        return
    arr = (L.AdamTensor * n)()
    for i, (p, g, m, v) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs)):
        assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32
        arr[i].param, arr[i].grad = p.data_ptr(), g.data_ptr()
        arr[i].exp_avg, arr[i].exp_avg_sq, arr[i].numel = m.data_ptr(), v.data_ptr(), p.numel()
    L.call("icg_adam_multi", ctypes.cast(arr, ctypes.c_void_p), n, float(lr), float(beta1), float(beta2), float(eps),
           int(step))
    bump_version(*params)


def nan_to_num_multi(tensors, nan=0.0, posinf=None, neginf=None):
    """torch.nan_to_num(t, nan, posinf, neginf, out=t) for every fp32 tensor of `tensors` in one launch per 64 tensors
    (icg_nan_to_num_multi; stylegan2_ada_pytorch/training/training_loop.py:511-515 does it per parameter gradient)."""
    import ctypes
    tensors = [t for t in tensors if t is not None and t.numel()]
    if not tensors:
        return
    _require_gpu(tensors[0])
    fmax = torch.finfo(torch.float32).max                 # torch.nan_to_num's defaults for None
    posinf, neginf = (fmax if posinf is None else posinf), (-fmax if neginf is None else neginf)
    arr = (L.F32Buffer * len(tensors))()
    for i, t in enumerate(tensors):
        assert t.dtype == torch.float32 and t.is_contiguous(), "nan_to_num_multi: contiguous fp32 tensors"
        arr[i].data, arr[i].numel = t.data_ptr(), t.numel()
    L.call("icg_nan_to_num_multi", ctypes.cast(arr, ctypes.c_void_p), len(tensors), float(nan), float(posinf), float(neginf))
    bump_version(*tensors)


def sg2_weight_prep_multi(items):
    """icg_sg2_weight_prep_multi: the per-pass weight preparation of StyleGAN2 layers (gain or fp16 pre-normalisation, cast, the
    gather-convolution layout and its adjoint, the demodulation table) for ALL given layers in two launches.  items: dicts with the
    tensors w [O][I][R][R], w_fwd, w_adj / None, wsq / None, wscale, warg / None and prenorm, gain, flip."""
    import ctypes
    if not items:
        return
    _require_gpu(items[0]["w"])
    arr = (L.Sg2Weight * len(items))()
    for i, it in enumerate(items):
        w = it["w"]
        assert w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 and w.shape[2] == w.shape[3]
        a = arr[i]
        a.w, a.w_fwd, a.wscale = w.data_ptr(), it["w_fwd"].data_ptr(), it["wscale"].data_ptr()
        a.w_adj = it["w_adj"].data_ptr() if it["w_adj"] is not None else None
        a.wsq = it["wsq"].data_ptr() if it["wsq"] is not None else None
        a.warg = it["warg"].data_ptr() if it["warg"] is not None else None
        a.O, a.I, a.R, a.prenorm = int(w.shape[0]), int(w.shape[1]), int(w.shape[2]), int(bool(it["prenorm"]))
        a.gain, a.flip, a.dtype = float(it["gain"]), int(bool(it["flip"])), 1 if it["w_fwd"].dtype == torch.float16 else 0
    L.call("icg_sg2_weight_prep_multi", ctypes.cast(arr, ctypes.c_void_p), len(items))


def ema_multi(targets, sources, decay):
    import ctypes
    n = len(targets)
    if n == 0:
        return
    arr = (L.EmaTensor * n)()
    for i, (t, s) in enumerate(zip(targets, sources)):
        assert t.is_contiguous() and s.is_contiguous() and t.dtype == torch.float32 and s.dtype == torch.float32
        arr[i].target, arr[i].source, arr[i].numel = t.data_ptr(), s.data_ptr(), t.numel()
    L.call("icg_ema_multi", ctypes.cast(arr, ctypes.c_void_p), n, float(decay))

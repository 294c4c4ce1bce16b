"""MI355X-native layer set for IC-GAN's BigGAN backbone.

Mirrors the public surface of the reference's ``BigGAN_PyTorch/layers.py`` (class names, constructor
arguments, parameter / buffer names -> identical ``state_dict`` layout) but every forward is a short
sequence of fused HIP kernels (``ic_gan_amd.ops``):

  reference op graph (layers.py:542-552)            here
  ---------------------------------------------     ---------------------------------------------------
  ccbn -> ReLU -> F.interpolate -> conv3x3          1 statistics pass + 1 conv whose operand loader applies
                                                    the per-sample affine, ReLU and the upsample index map
  conv_sc on the upsampled input, h + x             1x1 conv at LOW resolution, added (upsample-on-read) in
                                                    the epilogue of conv2
  SN.W_ per layer: 8 tiny ATen kernels              icg_sn_forward: W/sigma emitted in the two MFMA layouts

Unsupported reference options (never used by a shipped config) raise NotImplementedError instead of
silently taking another path: num_svs/num_itrs != 1, mybn, the reference's `cross_replica` nn.BatchNorm2d
variant (use `sync_bn=True`: cross-replica statistics over RCCL with the same buffers), norm_style != 'bn'.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter as P

from . import ops


# ------------------------------------------------------------------------------------------------
# Spectral norm  (reference layers.py:66-112)
# ------------------------------------------------------------------------------------------------
class SN(object):
    """Mixin holding the power-iteration buffers ``u0`` / ``sv0`` (same names as the reference)."""

    def _sn_init(self, num_svs, num_itrs, num_outputs, transpose=False, eps=1e-12):
        if num_svs != 1 or num_itrs != 1 or transpose:
            raise NotImplementedError("ic_gan_amd implements the shipped setting num_svs = num_itrs = 1")
        self.num_itrs, self.num_svs, self.transpose, self.eps = num_itrs, num_svs, transpose, eps
        self.register_buffer("u0", torch.randn(1, num_outputs))
        self.register_buffer("sv0", torch.ones(1))
        self._sn_eval = None       # (key, SNState): eval-mode cache of W/sigma (see _sn_eval_key)
        self._sn_flags = {}        # grad mode -> (need_dgrad, upsample, downsample) of the last call (see sn_prefetch)
        self._sn_ready = None      # (flags, SNState) computed ahead by sn_prefetch, consumed by the next sn_state()

    @property
    def u(self):
        return [self.u0]

    @property
    def sv(self):
        return [self.sv0]

    def sn_state(self, need_dgrad=None, upsample=False, downsample=False, _record=True, winograd=False) -> ops.SNState:
        """One power-iteration step (in place on u0/sv0 in training mode) + W/sigma in kernel layouts."""
        if need_dgrad is None:
            need_dgrad = torch.is_grad_enabled()
        flags = (bool(need_dgrad), bool(upsample), bool(downsample), int(winograd))
        if _record:
            self._sn_flags[torch.is_grad_enabled()] = flags
        ready, self._sn_ready = self._sn_ready, None
        if ready is not None:
            if ready[0] != flags:
                raise RuntimeError("spectral-norm state was prefetched for layouts %r but is consumed with %r" % (ready[0], flags))
            return ready[1]
        key = self._sn_eval_key(flags)
        if key is not None and self._sn_eval is not None and self._sn_eval[0] == key:
            return self._sn_eval[1]
        st = ops.sn_prepare(self.weight, self.u0, self.sv0, self.eps, self.training, need_dgrad, upsample, downsample,
                            winograd)
        if key is not None:
            self._sn_eval = (key, st)
        return st

    def _sn_eval_key(self, flags):
        """In eval mode without autograd W/sigma is a pure function of (weight, u0): it is computed once and reused until
        either tensor is written to (the reference recomputes it on every call; same values).  None = not cacheable.

        OPT-IN per network (enable_sn_eval_cache): the key relies on autograd version counters, which writes through
        `.data` aliases (the reference's utils.ema.update, any p.data.copy_()) do not bump.  A network whose weights
        may be written that way (G_ema under the reference's trainer) must not cache; the owners of frozen weights
        (inference.load_model_inference, inference.GraphedGenerator, bench.py's sampling workload) opt in."""
        if self.training or torch.is_grad_enabled() or not SN_EVAL_CACHE or not getattr(self, "_sn_cache_ok", False):
            return None
        return (flags, self.weight._version, self.u0._version, self.weight.data_ptr(), self.u0.data_ptr())

    def _sn_invalidate(self):
        self._sn_eval = None
        self._sn_ready = None

    def W_(self):
        """Spectrally normalised weight in the parameter layout (debug / API parity; not on the hot path)."""
        st = self.sn_state(False, _record=False)
        w = self.weight
        if w.dim() == 4:
            return st.w_ohwi.view(w.shape[0], w.shape[2], w.shape[3], w.shape[1]).permute(0, 3, 1, 2).contiguous()
        return st.w_ohwi.view_as(w)


SN_EVAL_CACHE = True      # global kill switch for the eval-mode W/sigma cache (GraphedGenerator turns it off while capturing)


def enable_sn_eval_cache(module, on=True):
    """Opt a network's spectral-norm layers in to (or out of) the eval-mode W/sigma cache.  Only for networks whose
    weights are frozen or written exclusively by this package's kernels (FusedAdam, utils.ema, the power iteration: they
    bump the version counters the cache key reads)."""
    for m in module.modules():
        if isinstance(m, SN):
            m._sn_cache_ok = bool(on)
            m._sn_invalidate()
    return module


def invalidate_sn_cache(module):
    """Drop every cached / prefetched W/sigma under `module` (call after writing weights or u0 through `.data`)."""
    for m in module.modules():
        if isinstance(m, SN):
            m._sn_invalidate()
    return module


def sn_drop_prefetched(modules):
    """Forget spectral-norm states that sn_prefetch computed and the forward did not consume (aborted forward)."""
    for m in modules:
        m._sn_ready = None


def sn_prefetch(modules):
    """Run the power iteration + weight normalisation of all `modules` (SN layers a network is about to call, in any
    order) as ONE batched pass instead of 5-6 launches per layer.  A layer takes part once the layouts it needs are known
    from its previous call in the same grad mode; the others keep the per-layer path.  Same kernels' arithmetic, so the
    result is bit-identical; each prefetched state must be consumed by exactly one sn_state() call of this forward."""
    mode = torch.is_grad_enabled()
    groups = {}
    for m in modules:
        if m._sn_ready is not None:
            # an earlier forward aborted (OOM, shape error, KeyboardInterrupt) or skipped this layer: the state is stale.
            # Its power-iteration step has already been applied to u0/sv0 (one extra iteration, harmless for the
            # estimate); drop it and carry on instead of bricking the module.
            warnings.warn("%s: dropping a prefetched spectral-norm state that the previous forward never consumed"
                          % type(m).__name__, RuntimeWarning, stacklevel=2)
            m._sn_ready = None
        if mode in m._sn_flags:
            key = m._sn_eval_key(m._sn_flags[mode])
            if key is not None and m._sn_eval is not None and m._sn_eval[0] == key:
                continue                                   # eval-mode cache is current: nothing to compute
            groups.setdefault((float(m.eps), bool(m.training)), []).append(m)
    for (eps, training), ms in groups.items():
        items = [(m.weight, m.u0, m.sv0) + m._sn_flags[mode] for m in ms]
        states = ops.sn_prepare_many(items, eps, training)
        ops.sn_group(states, [m.weight for m in ms])          # under autograd: one spectral-norm backward per group of layers
        for m, st in zip(ms, states):
            m._sn_ready = (m._sn_flags[mode], st)
            key = m._sn_eval_key(m._sn_flags[mode])
            if key is not None:
                m._sn_eval = (key, st)


class SNConv2d(nn.Conv2d, SN):
    def train(self, mode=True):
        self._sn_invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._sn_invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 num_svs=1, num_itrs=1, eps=1e-12):
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        k = self.kernel_size[0]
        if self.kernel_size != (k, k) or k not in (1, 3) or self.stride != (1, 1) or self.padding != (k // 2, k // 2) \
                or self.dilation != (1, 1) or groups != 1:
            raise NotImplementedError("ic_gan_amd.SNConv2d: 1x1 / 3x3, stride 1, 'same' padding only")
        self._sn_init(num_svs, num_itrs, out_channels, eps=eps)

    def forward(self, x, **fuse):
        # a 3x3 conv fed by a nearest x2 upsample runs in 4-phase form (needs channel counts the vector loader takes)
        phase = bool(fuse.get("upsample")) and self.kernel_size == (3, 3) and self.in_channels % 4 == 0 \
            and self.out_channels % 4 == 0 and fuse.get("residual") is None
        down = bool(fuse.get("downsample"))
        wino = 0
        if self.kernel_size == (3, 3) and not phase and not down and not fuse.get("upsample"):
            wino = ops.winograd_applies(self.in_channels, self.out_channels, x.shape[2], x.shape[3], x.shape[0])
        elif self.kernel_size == (3, 3) and (phase or down):
            # resample-fused layer: 25-plane F(4x4,3x3) domain where it pays (per direction, ops.RS_WINOGRAD_MIN_CHANNELS)
            up = 1 if phase else 0
            wino = ops.resample_winograd_applies(self.in_channels, self.out_channels, x.shape[2] << up, x.shape[3] << up,
                                                 x.shape[0])
        return ops.fused_conv(x, self.weight, self.bias, self.sn_state(upsample=phase, downsample=down, winograd=wino),
                              **fuse)


class SNLinear(nn.Linear, SN):
    def train(self, mode=True):
        self._sn_invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._sn_invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def __init__(self, in_features, out_features, bias=True, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Linear.__init__(self, in_features, out_features, bias)
        self._sn_init(num_svs, num_itrs, out_features, eps=eps)

    def forward(self, x):
        lead = x.shape[:-1]
        out = ops.linear(x.reshape(-1, x.shape[-1]), self.weight, self.bias, self.sn_state())
        return out.view(*lead, self.out_features)


class SNEmbedding(nn.Embedding, SN):
    def train(self, mode=True):
        self._sn_invalidate()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._sn_invalidate()
        return super()._load_from_state_dict(*args, **kwargs)

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, max_norm=None, norm_type=2,
                 scale_grad_by_freq=False, sparse=False, _weight=None, num_svs=1, num_itrs=1, eps=1e-12):
        nn.Embedding.__init__(self, num_embeddings, embedding_dim, padding_idx, max_norm, norm_type,
                              scale_grad_by_freq, sparse, _weight)
        self._sn_init(num_svs, num_itrs, num_embeddings, eps=eps)

    def forward(self, x):
        st = self.sn_state(False)
        return ops.SNEmbeddingFn.apply(x, self.weight if st.handle is None else st.handle, st)


class identity(nn.Module):
    def forward(self, input):
        return input


# ------------------------------------------------------------------------------------------------
# Normalisation  (reference layers.py:359-503)
# ------------------------------------------------------------------------------------------------
def _check_bn_variant(cross_replica, mybn, norm_style="bn"):
    if mybn or norm_style != "bn":
        raise NotImplementedError("ic_gan_amd: only norm_style='bn' with mybn=False (all shipped configs)")
    if cross_replica:
        raise NotImplementedError(
            "the reference's cross_replica=True builds a plain nn.BatchNorm2d (SURVEY F2); use sync_bn=True for "
            "cross-replica statistics over RCCL")


class ccbn(nn.Module):
    """Class / instance-conditional BatchNorm.  The normalise + affine (+ReLU, + upsample) is applied inside the
    next convolution; this module owns the parameters / running statistics and produces the per-sample
    gain and bias."""

    def __init__(self, output_size, input_size, which_linear, eps=1e-5, momentum=0.1, cross_replica=False,
                 mybn=False, norm_style="bn", sync_bn=False):
        super().__init__()
        _check_bn_variant(cross_replica, mybn, norm_style)
        self.output_size, self.input_size = output_size, input_size
        self.gain = which_linear(input_size, output_size)
        self.bias = which_linear(input_size, output_size)
        self.eps, self.momentum = eps, momentum
        self.cross_replica, self.mybn, self.norm_style, self.sync_bn = cross_replica, mybn, norm_style, sync_bn
        self.register_buffer("stored_mean", torch.zeros(output_size))
        self.register_buffer("stored_var", torch.ones(output_size))

    def affine(self, y):
        """-> (gain(y), bias(y)), each [B, C]; the '1 +' of the reference is folded in as gain_offset."""
        return self.gain(y), self.bias(y)

    def bn_opt(self) -> ops.BNOpt:
        # the reference hard-codes momentum 0.1 in this branch (layers.py:412-421)
        return ops.BNOpt(self.stored_mean, self.stored_var, self.eps, 0.1, self.training, 1.0,
                         True if self.sync_bn else None)

    def forward(self, x, y):
        gain, beta = self.affine(y)
        return ops.norm_act(x, self.bn_opt(), gain, beta, relu=False)

    def extra_repr(self):
        return f"out: {self.output_size}, in: {self.input_size}, sync_bn={self.sync_bn}"


def ccbn_affine_pair(bn1, bn2, y):
    """-> (gain1, bias1, gain2, bias2) of a block's two conditional BatchNorms for the same y: one grouped launch per direction
    (ops.CcbnAffineFn) when the four projections are bias-free SNLinear layers, else the four layers one by one."""
    mods = (bn1.gain, bn1.bias, bn2.gain, bn2.bias)
    # (from 16 rows: the grouped kernels tile 16 rows per wave, and below that the per-layer entry points take other kernels)
    if ops.GROUPED_CCBN and y.dim() == 2 and y.shape[0] >= 16 and all(isinstance(m, SNLinear) and m.bias is None
                                                                     and m.out_features % 4 == 0 for m in mods):
        sns = tuple(m.sn_state() for m in mods)
        ws = [m.weight if st.handle is None else st.handle for m, st in zip(mods, sns)]
        return ops.CcbnAffineFn.apply(y, sns, *ws)
    return bn1.affine(y) + bn2.affine(y)


class bn(nn.Module):
    """Plain affine BatchNorm (generator output layer)."""

    def __init__(self, output_size, eps=1e-5, momentum=0.1, cross_replica=False, mybn=False, sync_bn=False, **kwargs):
        super().__init__()
        _check_bn_variant(cross_replica, mybn)
        self.output_size, self.eps, self.momentum = output_size, eps, momentum
        self.cross_replica, self.mybn, self.sync_bn = cross_replica, mybn, sync_bn
        self.register_buffer("stored_mean", torch.zeros(output_size))
        self.register_buffer("stored_var", torch.ones(output_size))
        self.gain = P(torch.ones(output_size), requires_grad=True)
        self.bias = P(torch.zeros(output_size), requires_grad=True)

    def bn_opt(self) -> ops.BNOpt:
        return ops.BNOpt(self.stored_mean, self.stored_var, self.eps, self.momentum, self.training, 0.0,
                         True if self.sync_bn else None)

    def forward(self, x, y=None):
        return ops.norm_act(x, self.bn_opt(), self.gain, self.bias, relu=False)


# ------------------------------------------------------------------------------------------------
# Self-attention  (reference layers.py:206-244)
# ------------------------------------------------------------------------------------------------
def _plain_1x1(*convs):
    return all(isinstance(m, SNConv2d) and m.kernel_size == (1, 1) and m.bias is None for m in convs)


class Attention(nn.Module):
    def __init__(self, ch, which_conv=SNConv2d, name="attention"):
        super().__init__()
        self.ch, self.which_conv = ch, which_conv
        self.theta = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.phi = which_conv(ch, ch // 8, kernel_size=1, padding=0, bias=False)
        self.g = which_conv(ch, ch // 2, kernel_size=1, padding=0, bias=False)
        self.o = which_conv(ch // 2, ch, kernel_size=1, padding=0, bias=False)
        self.gamma = P(torch.tensor(0.0), requires_grad=True)

    def forward(self, x, y=None):
        # x has four consumers (theta, phi, g and the residual path).  Each projection hands x on to the next consumer
        # (`chain=True`: second output = x itself), so that in the backward pass the gradients of x arrive one after the
        # other and every projection adds the running sum in the epilogue of its own data-gradient GEMM: no elementwise
        # gradient-accumulation passes over [B, C, 64, 64] (three per block and backward pass, 1.2 GB of traffic each at cfg3)
        if ops.attn_projections_apply(x, self.theta.out_channels, self.g.out_channels) and _plain_1x1(self.theta, self.phi, self.g):
            # the three projections as ONE 1x1 convolution with stacked weights + one split / max-pool pass (ops.AttnProjFn)
            sns = tuple(m.sn_state() for m in (self.theta, self.phi, self.g))
            ws = [m.weight if st.handle is None else st.handle for m, st in zip((self.theta, self.phi, self.g), sns)]
            theta, phi, g, x = ops.AttnProjFn.apply(x, ws[0], ws[1], ws[2], sns)
        else:
            theta, x = self.theta(x, chain=True)
            phi, x = self.phi(x, chain=True)
            g, x = self.g(x, chain=True)
            phi, g = ops.MaxPool2Fn.apply(phi), ops.MaxPool2Fn.apply(g)
        a = ops.AttnCoreFn.apply(theta, phi, g)
        if ops.FUSED_ATTENTION_OUTPUT and _plain_1x1(self.o):
            # gamma folded into the output projection's weight, x added in its epilogue (ops.AttnOutFn).  The weight gradient needs
            # W / sigma in the [Cin][Cout] layout, so that layout is prepared whenever a gradient can be asked for
            st = self.o.sn_state()
            return ops.AttnOutFn.apply(a, x, self.o.weight if st.handle is None else st.handle, self.gamma, st)
        return ops.ScaleAddFn.apply(self.gamma, self.o(a), x)


# ------------------------------------------------------------------------------------------------
# Residual blocks  (reference layers.py:512-613)
# ------------------------------------------------------------------------------------------------
def _is_relu(act):
    return isinstance(act, nn.ReLU) or act is F.relu or act is torch.relu


class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=SNConv2d, which_bn=bn, activation=None, upsample=None):
        super().__init__()
        if not _is_relu(activation):
            raise NotImplementedError("ic_gan_amd.GBlock fuses ReLU; other G_nl settings are not implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.which_conv, self.which_bn = which_conv, which_bn
        self.activation = activation
        self.conv1 = which_conv(in_channels, out_channels)
        self.conv2 = which_conv(out_channels, out_channels)
        self.learnable_sc = in_channels != out_channels or upsample
        if self.learnable_sc:
            self.conv_sc = which_conv(in_channels, out_channels, kernel_size=1, padding=0)
        self.bn1 = which_bn(in_channels)
        self.bn2 = which_bn(out_channels)
        self.upsample = upsample   # truthiness only: nearest x2 is folded into conv1 / the residual read

    def forward(self, x, y):
        up = bool(self.upsample)
        o1, o2 = self.bn1.bn_opt(), self.bn2.bn_opt()
        if ops._sync_enabled(o1):
            return self._forward_sync(x, y, up, o1, o2)
        if isinstance(self.bn1, ccbn) and isinstance(self.bn2, ccbn):
            g1, b1, g2, b2 = ccbn_affine_pair(self.bn1, self.bn2, y)
        else:
            g1, b1 = self.bn1.affine(y) if isinstance(self.bn1, ccbn) else (self.bn1.gain, self.bn1.bias)
            g2, b2 = self.bn2.affine(y) if isinstance(self.bn2, ccbn) else (self.bn2.gain, self.bn2.bias)
        # 1x1 shortcut commutes with nearest upsampling: run it at the input resolution.  x feeds the shortcut and the main path:
        # the shortcut hands x on (`chain=True`, see Attention.forward), so the main path's gradient of x is added in the
        # epilogue of the shortcut's data-gradient GEMM instead of by an elementwise pass
        if self.learnable_sc:
            sc, x = self.conv_sc(x, chain=True)
        else:
            sc = x
        h = self.conv1(x, relu=True, upsample=up, bn=o1, gain=g1, beta=b1)
        return self.conv2(h, relu=True, bn=o2, gain=g2, beta=b2, residual=sc, res_up=up)

    def _forward_sync(self, x, y, up, o1, o2):
        """Cross-replica BN: same arithmetic, launch order chosen so that each statistic all-reduce (latency-bound, on
        RCCL's stream) has independent kernels to hide behind -- bn1's behind the four conditioning projections, bn2's
        behind the 1x1 shortcut convolution."""
        st1 = ops.bn_stats_begin(x, o1)
        if isinstance(self.bn1, ccbn) and isinstance(self.bn2, ccbn):
            g1, b1, g2, b2 = ccbn_affine_pair(self.bn1, self.bn2, y)
        else:
            g1, b1 = self.bn1.affine(y) if isinstance(self.bn1, ccbn) else (self.bn1.gain, self.bn1.bias)
            g2, b2 = self.bn2.affine(y) if isinstance(self.bn2, ccbn) else (self.bn2.gain, self.bn2.bias)
        h = self.conv1(x, relu=True, upsample=up, bn=o1, gain=g1, beta=b1, bn_stats=st1)
        st2 = ops.bn_stats_begin(h, o2)
        sc = self.conv_sc(x) if self.learnable_sc else x          # (launch order matters here: no gradient chain)
        return self.conv2(h, relu=True, bn=o2, gain=g2, beta=b2, residual=sc, res_up=up, bn_stats=st2)


class DBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=SNConv2d, wide=True, preactivation=False,
                 activation=None, downsample=None):
        super().__init__()
        if not _is_relu(activation):
            raise NotImplementedError("ic_gan_amd.DBlock fuses ReLU; other D_nl settings are not implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = out_channels if wide else in_channels
        self.which_conv, self.preactivation, self.activation = which_conv, preactivation, activation
        self.downsample = downsample   # truthiness only: 2x2 average pooling kernel
        self.conv1 = which_conv(in_channels, self.hidden_channels)
        self.conv2 = which_conv(self.hidden_channels, out_channels)
        self.learnable_sc = True if (in_channels != out_channels) or downsample else False
        if self.learnable_sc:
            self.conv_sc = which_conv(in_channels, out_channels, kernel_size=1, padding=0)

    def forward(self, x):
        if self.downsample:
            # avg-pool and the 1x1 shortcut are both linear: pool first (4x fewer MACs) for either block kind.  The pooling hands
            # x on to the main path (gradient chain, see Attention.forward): conv1's gradient of x is added inside the pooling
            # backward kernel instead of by an elementwise pass
            s, x = ops.AvgPool2Fn.apply(x, None, True)
            h = self.conv1(x, relu=bool(self.preactivation))
            if self.learnable_sc:
                s = self.conv_sc(s)
            if self.conv2.in_channels % 4 == 0 and self.conv2.out_channels % 4 == 0:
                # conv2 + AvgPool2d + residual add as ONE 4x4/stride-2 conv at the pooled resolution
                return self.conv2(h, relu=True, downsample=True, residual=s)
            h = self.conv2(h, relu=True)
            return ops.AvgPool2Fn.apply(h, s)
        h = self.conv1(x, relu=bool(self.preactivation))
        s = self.conv_sc(x) if self.learnable_sc else x
        return self.conv2(h, relu=True, residual=s)

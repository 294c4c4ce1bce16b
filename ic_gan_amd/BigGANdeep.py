"""BigGAN-deep generator / discriminator on the fused MI355X operators — module API of the reference's
`BigGAN_PyTorch/BigGANdeep.py` (GBlock 33-84, G_arch 87-130, Generator 133-391, DBlock 394-452, D_arch 455-497,
Discriminator 500-688, G_D 691-734): bottleneck residual blocks (1x1 -> 3x3 -> 3x3 -> 1x1 with a channel ratio of 4),
channel-dropping skip in G, channel-concatenating skip in D, `G_depth` / `D_depth` blocks per resolution.

In the reference this model is class-conditional only (`Generator.forward(z, y)` with y the shared embedding, no instance
features) and its `G_D` has no feature arguments, so `train_fns.GAN_training_function` (which always passes them) cannot
drive it.  That class-conditional form is kept bit-for-bit in API and state_dict (pinned by reference-generated
goldens, tests/test_biggan_deep.py).

INSTANCE CONDITIONING (BASELINE.json configs[4], "IC-GAN BigGANdeep") is added by analogy with BigGAN.py:350-358 (G's
`shared_feat` linear on the 2048-d instance feature, concatenated with the class embedding into the conditioning vector of
every ccbn), BigGAN.py:546-553,625-641 (D's projection head on class embedding ++ `linear_feat(feat)`) and BigGAN.py:655-711
(`G_D.forward(z, gy, feats_g, x, dy, feats, ...)`): pass `instance_cond=True` (and `class_cond=False` for the label-free
IC-GAN) and the module takes labels / features like ic_gan_amd.BigGAN and can be driven by train_fns.GAN_training_function.
The reference has no such model, so this part has NO oracle: parity unpinned (state_dict names follow BigGAN.py's:
`shared_feat.*`, `linear_feat.*`).

Fusions used (ic_gan_amd/ops.py): every convolution takes its preceding [ccbn affine ->] ReLU [-> nearest x2] in the operand
loader; G: conv2 after the upsample runs in 4-phase form, conv4 adds the (upsampled-on-read) skip in its epilogue; D: the
average pool after the last ReLU commutes with the 1x1 conv4 (both linear), so conv4 runs first and the pool adds the skip.
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers, ops
from .BigGAN import _attn_resolutions, _init_module_weights
from .optim import FusedAdam

# resolution -> (input multipliers, output multipliers)                                   (BigGANdeep.py:87-130)
_G_PLAN = {
    256: ((16, 16, 8, 8, 4, 2), (16, 8, 8, 4, 2, 1)),
    128: ((16, 16, 8, 4, 2), (16, 8, 4, 2, 1)),
    64: ((16, 16, 8, 4), (16, 8, 4, 2)),
    32: ((4, 4, 4), (4, 4, 4)),
}
# resolution -> (input mults, output mults, output resolutions, downsample flags)          (BigGANdeep.py:455-497)
_D_PLAN = {
    256: ((1, 2, 4, 8, 8, 16), (2, 4, 8, 8, 16, 16), (128, 64, 32, 16, 8, 4, 4), (1, 1, 1, 1, 1, 1, 0)),
    128: ((1, 2, 4, 8, 16), (2, 4, 8, 16, 16), (64, 32, 16, 8, 4, 4), (1, 1, 1, 1, 1, 0)),
    64: ((1, 2, 4, 8), (2, 4, 8, 16), (32, 16, 8, 4, 4), (1, 1, 1, 1, 0)),
    32: ((4, 4, 4), (4, 4, 4), (16, 16, 16, 16), (1, 1, 0, 0)),
}


def G_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    att = _attn_resolutions(attention)
    arch = {}
    for res, (cin, cout) in _G_PLAN.items():
        out_res = [8 << i for i in range(len(cout))]
        arch[res] = {"in_channels": [ch * m for m in cin], "out_channels": [ch * m for m in cout],
                     "upsample": [True] * len(cout), "resolution": out_res,
                     "attention": {r: (r in att) for r in out_res}}
    return arch


def D_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    att = _attn_resolutions(attention)
    arch = {}
    for res, (cin, cout, out_res, down) in _D_PLAN.items():
        arch[res] = {"in_channels": [ch * m for m in cin], "out_channels": [ch * m for m in cout],
                     "downsample": [bool(d) for d in down], "resolution": list(out_res),
                     "attention": {r: (r in att) for r in set(out_res)}}
    return arch


def _cond(bn_layer, y):
    return bn_layer.affine(y) if isinstance(bn_layer, layers.ccbn) else (bn_layer.gain, bn_layer.bias)


class GBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=layers.SNConv2d, which_bn=layers.bn, activation=None,
                 upsample=None, channel_ratio=4):
        super().__init__()
        if not layers._is_relu(activation):
            raise NotImplementedError("ic_gan_amd GBlock fuses ReLU; other G_nl settings are not implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = in_channels // channel_ratio
        self.which_conv, self.which_bn, self.activation = which_conv, which_bn, activation
        self.conv1 = which_conv(in_channels, self.hidden_channels, kernel_size=1, padding=0)
        self.conv2 = which_conv(self.hidden_channels, self.hidden_channels)
        self.conv3 = which_conv(self.hidden_channels, self.hidden_channels)
        self.conv4 = which_conv(self.hidden_channels, out_channels, kernel_size=1, padding=0)
        self.bn1 = which_bn(in_channels)
        self.bn2 = which_bn(self.hidden_channels)
        self.bn3 = which_bn(self.hidden_channels)
        self.bn4 = which_bn(self.hidden_channels)
        self.upsample = upsample       # truthiness only: nearest x2, folded into conv2's loader and conv4's skip read

    def forward(self, x, y):
        up = bool(self.upsample)
        (g1, b1), (g2, b2), (g3, b3), (g4, b4) = (_cond(m, y) for m in (self.bn1, self.bn2, self.bn3, self.bn4))
        h = self.conv1(x, relu=True, bn=self.bn1.bn_opt(), gain=g1, beta=b1)
        h = self.conv2(h, relu=True, upsample=up, bn=self.bn2.bn_opt(), gain=g2, beta=b2)
        h = self.conv3(h, relu=True, bn=self.bn3.bn_opt(), gain=g3, beta=b3)
        skip = x
        if self.in_channels != self.out_channels:          # drop channels (BigGANdeep.py:73-74)
            skip = x[:, : self.out_channels].contiguous(memory_format=torch.channels_last)
        return self.conv4(h, relu=True, bn=self.bn4.bn_opt(), gain=g4, beta=b4, residual=skip, res_up=up)


class DBlock(nn.Module):
    def __init__(self, in_channels, out_channels, which_conv=layers.SNConv2d, wide=True, preactivation=True,
                 activation=None, downsample=None, channel_ratio=4):
        super().__init__()
        if not layers._is_relu(activation):
            raise NotImplementedError("ic_gan_amd DBlock fuses ReLU; other D_nl settings are not implemented")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.hidden_channels = out_channels // channel_ratio
        self.which_conv, self.preactivation, self.activation = which_conv, preactivation, activation
        self.downsample = downsample   # truthiness only: 2x2 average pooling
        self.conv1 = which_conv(in_channels, self.hidden_channels, kernel_size=1, padding=0)
        self.conv2 = which_conv(self.hidden_channels, self.hidden_channels)
        self.conv3 = which_conv(self.hidden_channels, self.hidden_channels)
        self.conv4 = which_conv(self.hidden_channels, out_channels, kernel_size=1, padding=0)
        self.learnable_sc = in_channels != out_channels
        if self.learnable_sc:
            self.conv_sc = which_conv(in_channels, out_channels - in_channels, kernel_size=1, padding=0)

    def shortcut(self, x):
        if self.downsample:
            x = ops.AvgPool2Fn.apply(x, None)
        if self.learnable_sc:          # keep the input channels, learn the additional ones (BigGANdeep.py:431-436)
            x = torch.cat([x, self.conv_sc(x)], 1).contiguous(memory_format=torch.channels_last)
        return x

    def forward(self, x):
        h = self.conv1(x, relu=True)          # F.relu(x) in the reference: x itself stays un-activated for the skip
        h = self.conv2(h, relu=True)
        h = self.conv3(h, relu=True)
        s = self.shortcut(x)
        if self.downsample:
            # relu -> avgpool -> 1x1 conv == relu -> 1x1 conv -> avgpool (pool and 1x1 conv are linear and commute)
            return ops.AvgPool2Fn.apply(self.conv4(h, relu=True), s)
        return self.conv4(h, relu=True, residual=s)


class _OutputLayer(nn.Sequential):
    def forward(self, h):
        norm, _, conv = self[0], self[1], self[2]
        return conv(h, relu=True, bn=norm.bn_opt(), gain=norm.gain, beta=norm.bias)


class Generator(nn.Module):
    def __init__(self, G_ch=64, G_depth=2, dim_z=128, bottom_width=4, resolution=128, G_kernel_size=3, G_attn="64",
                 n_classes=1000, num_G_SVs=1, num_G_SV_itrs=1, G_shared=True, shared_dim=0, hier=False,
                 cross_replica=False, mybn=False, G_activation=nn.ReLU(inplace=False), G_lr=5e-5, G_B1=0.0, G_B2=0.999,
                 adam_eps=1e-8, BN_eps=1e-5, SN_eps=1e-12, G_mixed_precision=False, G_fp16=False, G_init="ortho",
                 skip_init=False, no_optim=False, G_param="SN", norm_style="bn", sync_bn=False, class_cond=True,
                 instance_cond=False, G_shared_feat=True, shared_dim_feat=2048, **kwargs):
        super().__init__()
        if G_param != "SN":
            raise NotImplementedError("ic_gan_amd.BigGANdeep.Generator: G_param='SN' only")
        if not class_cond and not instance_cond:
            raise NotImplementedError("BigGANdeep.Generator needs class and / or instance conditioning")
        if instance_cond and not (G_shared and hier):
            raise NotImplementedError("instance conditioning of BigGANdeep is defined for G_shared=True, hier=True (the "
                                      "shipped BigGAN-deep setting)")
        self.class_cond, self.instance_cond = class_cond, instance_cond
        self.G_shared_feat, self.shared_dim_feat = G_shared_feat, (shared_dim_feat if instance_cond else 0)
        if G_fp16 or G_mixed_precision:
            raise NotImplementedError("ic_gan_amd computes in fp32; fp16 modes are not implemented")
        self.ch, self.G_depth, self.dim_z, self.bottom_width = G_ch, G_depth, dim_z, bottom_width
        self.resolution, self.kernel_size, self.attention, self.n_classes = resolution, G_kernel_size, G_attn, n_classes
        self.G_shared = G_shared
        self.shared_dim = shared_dim if shared_dim > 0 else dim_z
        self.hier, self.cross_replica, self.mybn = hier, cross_replica, mybn
        self.activation, self.init, self.G_param, self.norm_style = G_activation, G_init, G_param, norm_style
        self.BN_eps, self.SN_eps, self.fp16 = BN_eps, SN_eps, G_fp16
        self.arch = G_arch(self.ch, self.attention)[resolution]

        sn_kw = dict(num_svs=num_G_SVs, num_itrs=num_G_SV_itrs, eps=self.SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn_kw)
        self.which_linear = functools.partial(layers.SNLinear, **sn_kw)
        self.which_embedding = nn.Embedding
        bn_linear = functools.partial(self.which_linear, bias=False) if self.G_shared else self.which_embedding
        # width of the conditioning vector y: class embedding (++ instance embedding, BigGAN.py:350-358 by analogy)
        self.cond_dim = (self.shared_dim if class_cond else 0) + self.shared_dim_feat
        self.which_bn = functools.partial(layers.ccbn, which_linear=bn_linear, cross_replica=self.cross_replica,
                                          mybn=self.mybn,
                                          input_size=(self.cond_dim + self.dim_z if self.G_shared else self.n_classes),
                                          norm_style=self.norm_style, eps=self.BN_eps, sync_bn=sync_bn)
        # `shared` always exists, as in BigGAN.py (identity when there is no class embedding): reference-style callers do
        # `G.shared(y)` unconditionally (train_fns.py:174, utils.py:1471-1474, G_D's class-conditional branch); an identity has no
        # parameters, so the state_dict of the instance-only model is unchanged
        self.shared = (self.which_embedding(n_classes, self.shared_dim) if G_shared else layers.identity()) if class_cond \
            else layers.identity()
        if instance_cond:
            self.shared_feat = self.which_linear(2048, self.shared_dim_feat) if G_shared_feat else layers.identity()
        self.linear = self.which_linear(self.dim_z + self.cond_dim,
                                        self.arch["in_channels"][0] * (self.bottom_width ** 2))
        stages = []
        for i in range(len(self.arch["out_channels"])):
            # the reference appends ONE list per depth block (BigGANdeep.py:284-305): state_dict keys are
            # blocks.{i * G_depth + g}.0.*, the attention layer joins the list of the stage's last block
            group = [[GBlock(in_channels=self.arch["in_channels"][i],
                             out_channels=(self.arch["in_channels"][i] if g == 0 else self.arch["out_channels"][i]),
                             which_conv=self.which_conv, which_bn=self.which_bn, activation=self.activation,
                             upsample=(functools.partial(F.interpolate, scale_factor=2)
                                       if self.arch["upsample"][i] and g == self.G_depth - 1 else None))]
                     for g in range(self.G_depth)]
            if self.arch["attention"][self.arch["resolution"][i]]:
                print("Adding attention layer in G at resolution %d" % self.arch["resolution"][i])
                group[-1].append(layers.Attention(self.arch["out_channels"][i], self.which_conv))
            stages.extend(nn.ModuleList(b) for b in group)
        self.blocks = nn.ModuleList(stages)
        self.output_layer = _OutputLayer(
            layers.bn(self.arch["out_channels"][-1], cross_replica=self.cross_replica, mybn=self.mybn, sync_bn=sync_bn),
            self.activation, self.which_conv(self.arch["out_channels"][-1], 3))
        if not skip_init:
            self.init_weights()
        if no_optim:
            return
        self.lr, self.B1, self.B2, self.adam_eps = G_lr, G_B1, G_B2, adam_eps
        self.optim = FusedAdam(params=self.parameters(), lr=self.lr, betas=(self.B1, self.B2), weight_decay=0,
                               eps=self.adam_eps)

    def init_weights(self):
        self.param_count = _init_module_weights(self, self.init)
        print("Param count for G" "s initialized parameters: %d" % self.param_count)

    def get_condition_embeddings(self, cl=None, feat=None):
        """class embedding ++ instance embedding (BigGAN.py:350-358)."""
        parts = []
        if cl is not None:
            parts.append(self.shared(cl))
        if feat is not None:
            parts.append(self.shared_feat(feat))
        return torch.cat(parts, dim=-1)

    def forward(self, z, y=None, feats=None):
        """Reference form (instance_cond=False): z [B, dim_z], y [B, shared_dim] = the class embedding `self.shared(labels)`
        (BigGANdeep.py:375-391).  Instance-conditioned form: y = int64 labels [B] or None, feats [B, 2048] -> the embeddings
        are computed here, as in BigGAN.py:364-386."""
        sn_layers = [m for m in self.modules() if isinstance(m, layers.SN)]
        layers.sn_prefetch(sn_layers)
        try:
            if self.instance_cond:
                y = self.get_condition_embeddings(y if self.class_cond else None, feats)
            if self.hier:
                z = torch.cat([y, z], 1)
                y = z
            h = self.linear(z)
            h = h.view(h.size(0), -1, self.bottom_width, self.bottom_width)
            for stage in self.blocks:
                for block in stage:
                    h = block(h, y)
            return ops.TanhFn.apply(self.output_layer(h))
        finally:
            layers.sn_drop_prefetched(sn_layers)


class Discriminator(nn.Module):
    def __init__(self, D_ch=64, D_wide=True, D_depth=2, resolution=128, D_kernel_size=3, D_attn="64", n_classes=1000,
                 num_D_SVs=1, num_D_SV_itrs=1, D_activation=nn.ReLU(inplace=False), D_lr=2e-4, D_B1=0.0, D_B2=0.999,
                 adam_eps=1e-8, SN_eps=1e-12, output_dim=1, D_mixed_precision=False, D_fp16=False, D_init="ortho",
                 skip_init=False, D_param="SN", class_cond=True, instance_cond=False, instance_sz=2048, **kwargs):
        super().__init__()
        if D_param != "SN":
            raise NotImplementedError("ic_gan_amd.BigGANdeep.Discriminator: D_param='SN' only")
        self.class_cond, self.instance_cond = class_cond, instance_cond
        if D_fp16 or D_mixed_precision:
            raise NotImplementedError("ic_gan_amd computes in fp32; fp16 modes are not implemented")
        self.ch, self.D_wide, self.D_depth, self.resolution = D_ch, D_wide, D_depth, resolution
        self.kernel_size, self.attention, self.n_classes = D_kernel_size, D_attn, n_classes
        self.activation, self.init, self.D_param, self.SN_eps, self.fp16 = D_activation, D_init, D_param, SN_eps, D_fp16
        self.arch = D_arch(self.ch, self.attention)[resolution]
        sn_kw = dict(num_svs=num_D_SVs, num_itrs=num_D_SV_itrs, eps=self.SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn_kw)
        self.which_linear = functools.partial(layers.SNLinear, **sn_kw)
        self.which_embedding = functools.partial(layers.SNEmbedding, **sn_kw)
        self.input_conv = self.which_conv(3, self.arch["in_channels"][0])
        stages = []
        for i in range(len(self.arch["out_channels"])):
            stage = [DBlock(in_channels=(self.arch["in_channels"][i] if d == 0 else self.arch["out_channels"][i]),
                            out_channels=self.arch["out_channels"][i], which_conv=self.which_conv, wide=self.D_wide,
                            activation=self.activation, preactivation=True,
                            downsample=(nn.AvgPool2d(2) if self.arch["downsample"][i] and d == 0 else None))
                     for d in range(self.D_depth)]
            if self.arch["attention"][self.arch["resolution"][i]]:
                print("Adding attention layer in D at resolution %d" % self.arch["resolution"][i])
                stage.append(layers.Attention(self.arch["out_channels"][i], self.which_conv))
            stages.append(nn.ModuleList(stage))
        self.blocks = nn.ModuleList(stages)
        top = self.arch["out_channels"][-1]
        self.linear = self.which_linear(top, output_dim)
        # projection head: class embedding, or (BigGAN.py:546-553 by analogy) class embedding ++ linear_feat(instance feature)
        if class_cond and instance_cond:
            self.linear_feat = self.which_linear(instance_sz, top // 2)
            self.embed = self.which_embedding(self.n_classes, top // 2)
        elif instance_cond:
            self.linear_feat = self.which_linear(instance_sz, top)
        else:
            self.embed = self.which_embedding(self.n_classes, top)
        if not skip_init:
            self.init_weights()
        self.lr, self.B1, self.B2, self.adam_eps = D_lr, D_B1, D_B2, adam_eps
        self.optim = FusedAdam(params=self.parameters(), lr=self.lr, betas=(self.B1, self.B2), weight_decay=0,
                               eps=self.adam_eps)

    def init_weights(self):
        self.param_count = _init_module_weights(self, self.init)
        print("Param count for D" "s initialized parameters: %d" % self.param_count)

    def forward(self, x, y=None, feat=None):
        """x [N,3,R,R], y [N] int64 (or None), feat [N,2048] (instance-conditioned form) -> logits [N,1]
        (BigGANdeep.py:673-688; projection on class ++ instance embedding as BigGAN.py:625-641)."""
        skip = set()
        if (y is None or not self.class_cond) and hasattr(self, "embed"):
            skip |= {id(m) for m in self.embed.modules()}
        if feat is None and hasattr(self, "linear_feat"):
            skip |= {id(m) for m in self.linear_feat.modules()}
        sn_layers = [m for m in self.modules() if isinstance(m, layers.SN) and id(m) not in skip]
        layers.sn_prefetch(sn_layers)
        try:
            h = self.input_conv(x)
            for stage in self.blocks:
                for block in stage:
                    h = block(h)
            h = ops.ReluSumPoolFn.apply(h)
            out = self.linear(h)
            parts = []
            if self.class_cond and y is not None:
                parts.append(self.embed(y))
            if self.instance_cond and feat is not None:
                parts.append(self.linear_feat(feat))
            if not parts:
                return out
            proj = torch.cat(parts, dim=-1) if len(parts) > 1 else parts[0]
            return out + torch.sum(proj * h, 1, keepdim=True)
        finally:
            layers.sn_drop_prefetched(sn_layers)


class G_D(nn.Module):
    """BigGANdeep.py:691-734.  With an instance-conditioned generator the call signature is BigGAN.py's
    `forward(z, gy, feats_g, x, dy, feats, train_G, return_G_z, split_D, policy, DA)` (BigGAN.py:655-711), which is what
    train_fns.GAN_training_function passes positionally; otherwise the reference BigGAN-deep signature
    `forward(z, gy, x, dy, train_G, return_G_z, split_D)`."""

    def __init__(self, G, D, optimizer_G=None, optimizer_D=None):
        super().__init__()
        self.G, self.D = G, D
        self.optimizer_G, self.optimizer_D = optimizer_G, optimizer_D

    def _bare(self, m):
        return m.module if hasattr(m, "module") else m

    def generate(self, z, gy, feats_g=None):
        """the generator call of `forward` (what train_fns.PREFETCH_NEXT_STEP issues ahead, under no_grad)"""
        if getattr(self._bare(self.G), "instance_cond", False):
            return self.G(z, gy, feats_g)
        return self.G(z, self._bare(self.G).shared(gy))

    def forward(self, z, gy, *args, **kwargs):
        if getattr(self._bare(self.G), "instance_cond", False):
            return self._forward_ic(z, gy, *args, **kwargs)
        return self._forward_cc(z, gy, *args, **kwargs)

    def _forward_cc(self, z, gy, x=None, dy=None, train_G=False, return_G_z=False, split_D=False, G_z=None):
        if G_z is None:
            with torch.set_grad_enabled(train_G):
                G_z = self.generate(z, gy)
        else:
            assert not train_G and not G_z.requires_grad
        if split_D:
            D_fake = self.D(G_z, gy)
            if x is not None:
                return D_fake, self.D(x, dy)
            return (D_fake, G_z) if return_G_z else D_fake
        D_input = torch.cat([G_z, x.contiguous(memory_format=torch.channels_last)], 0) if x is not None else G_z
        D_class = torch.cat([gy, dy], 0) if dy is not None else gy
        D_out = self.D(D_input, D_class)
        if x is not None:
            return torch.split(D_out, [G_z.shape[0], x.shape[0]])
        return (D_out, G_z) if return_G_z else D_out

    def _forward_ic(self, z, gy, feats_g=None, x=None, dy=None, feats=None, train_G=False, return_G_z=False, split_D=False,
                    policy=False, DA=False, G_z=None):
        """`G_z`: the generator's output for these inputs computed ahead under no_grad (train_fns.PREFETCH_NEXT_STEP; BigGAN.G_D.forward)"""
        if DA:
            raise NotImplementedError("DiffAugment is disabled in every shipped IC-GAN config (SURVEY 2.1)")
        if G_z is None:
            with torch.set_grad_enabled(train_G):
                G_z = self.generate(z, gy, feats_g)
        else:
            assert not train_G and not G_z.requires_grad
        if split_D:
            D_fake = self.D(G_z, gy, feats_g)
            if x is not None:
                return D_fake, self.D(x, dy, feats)
            return (D_fake, G_z) if return_G_z else D_fake
        D_input = torch.cat([G_z, x.contiguous(memory_format=torch.channels_last)], 0) if x is not None else G_z
        D_class = (torch.cat([gy, dy], 0) if dy is not None else gy) if gy is not None else None
        D_feats = (torch.cat([feats_g, feats], 0) if feats is not None else feats_g) if feats_g is not None else None
        D_out = self.D(D_input, D_class, D_feats)
        if x is not None:
            return torch.split(D_out, [G_z.shape[0], x.shape[0]])
        return (D_out, G_z) if return_G_z else D_out

"""IC-GAN BigGAN Generator / Discriminator / G_D on the MI355X kernel set.

Drop-in for the model module the reference selects with ``__import__(config["model"])``
(BigGAN_PyTorch/trainer.py:122): same constructor keywords, ``forward`` signatures, attributes read by
callers (``dim_z``, ``shared``, ``fp16``, ``optim``) and — checked by tests/test_host_logic_cpu.py::test_state_dict_contract —
the same ``state_dict()`` names and shapes as BigGAN_PyTorch/BigGAN.py, so checkpoints written by either
implementation load into the other with ``strict=True``.

The network description is data (``_G_PLAN`` / ``_D_PLAN``: channel multipliers per resolution) instead of
the reference's per-resolution dict literals; the forward passes are sequences of fused kernels
(see ic_gan_amd/layers.py).
"""
from __future__ import annotations

import functools

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from . import layers, ops
from .optim import FusedAdam

# resolution -> (input multipliers, output multipliers) of the generator blocks      (BigGAN.py:32-85)
_G_PLAN = {
    512: ((16, 16, 8, 8, 4, 2, 1), (16, 8, 8, 4, 2, 1, 1)),
    256: ((16, 16, 8, 8, 4, 2), (16, 8, 8, 4, 2, 1)),
    128: ((16, 16, 8, 4, 2), (16, 8, 4, 2, 1)),
    64: ((16, 16, 8, 4), (16, 8, 4, 2)),
    32: ((4, 4, 4), (4, 4, 4)),
}
# resolution -> (input mults after the RGB stem, output mults, output resolutions, downsample flags)  (BigGAN.py:390-432)
_D_PLAN = {
    256: ((1, 2, 4, 8, 8, 16), (1, 2, 4, 8, 8, 16, 16), (128, 64, 32, 16, 8, 4, 4), (1, 1, 1, 1, 1, 1, 0)),
    128: ((1, 2, 4, 8, 16), (1, 2, 4, 8, 16, 16), (64, 32, 16, 8, 4, 4), (1, 1, 1, 1, 1, 0)),
    64: ((1, 2, 4, 8), (1, 2, 4, 8, 16), (32, 16, 8, 4, 4), (1, 1, 1, 1, 0)),
    32: ((4, 4, 4), (4, 4, 4, 4), (16, 16, 16, 16), (1, 1, 0, 0)),
}


def _attn_resolutions(spec):
    return {int(tok) for tok in str(spec).split("_") if tok}


def G_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    att = _attn_resolutions(attention)
    arch = {}
    for res, (cin, cout) in _G_PLAN.items():
        n = len(cout)
        out_res = [8 << i for i in range(n)]
        arch[res] = {
            "in_channels": [ch * m for m in cin],
            "out_channels": [ch * m for m in cout],
            "upsample": [True] * n,
            "resolution": out_res,
            "attention": {r: (r in att) for r in out_res},
        }
    return arch


def D_arch(ch=64, attention="64", ksize="333333", dilation="111111"):
    att = _attn_resolutions(attention)
    arch = {}
    for res, (cin, cout, out_res, down) in _D_PLAN.items():
        arch[res] = {
            "in_channels": [3] + [ch * m for m in cin],
            "out_channels": [ch * m for m in cout],
            "downsample": [bool(d) for d in down],
            "resolution": list(out_res),
            "attention": {r: (r in att) for r in set(out_res)},
        }
    return arch


def _init_module_weights(net, style):
    """Reference Generator.init_weights / Discriminator.init_weights (BigGAN.py:324-345, 594-615)."""
    count = 0
    for module in net.modules():
        if isinstance(module, (nn.Conv2d, nn.Linear, nn.Embedding)):
            if style == "ortho":
                init.orthogonal_(module.weight)
            elif style == "N02":
                init.normal_(module.weight, 0, 0.02)
            elif style in ("glorot", "xavier"):
                init.xavier_uniform_(module.weight)
            else:
                print("Init style not recognized...")
            count += sum(p.data.nelement() for p in module.parameters())
    return count


class _OutputLayer(nn.Sequential):
    """bn -> ReLU -> conv (state_dict keys ``output_layer.0.*`` / ``output_layer.2.*``) executed as one fused
    normalise+ReLU+conv kernel sequence."""

    def forward(self, h):
        norm, _, conv = self[0], self[1], self[2]
        return conv(h, relu=True, bn=norm.bn_opt(), gain=norm.gain, beta=norm.bias)


class Generator(nn.Module):
    def __init__(self, G_ch=64, dim_z=128, bottom_width=4, resolution=128, G_kernel_size=3, G_attn="64",
                 n_classes=1000, num_G_SVs=1, num_G_SV_itrs=1, G_shared=True, shared_dim=0, hier=False,
                 cross_replica=False, mybn=False, G_activation=nn.ReLU(inplace=False), G_lr=5e-5, G_B1=0.0,
                 G_B2=0.999, adam_eps=1e-8, BN_eps=1e-5, SN_eps=1e-12, G_mixed_precision=False, G_fp16=False,
                 G_init="ortho", skip_init=False, no_optim=False, G_param="SN", norm_style="bn", class_cond=True,
                 embedded_optimizer=True, instance_cond=False, G_shared_feat=True, shared_dim_feat=2048,
                 sync_bn=False, **kwargs):
        super().__init__()
        if G_param != "SN":
            raise NotImplementedError("ic_gan_amd.Generator: G_param='SN' only (every shipped config)")
        if G_fp16 or G_mixed_precision:
            raise NotImplementedError("ic_gan_amd computes in fp32 (exact fp32 MFMA); fp16 modes are not implemented")
        # tolerate the reference's own spelling slip (trainer.py passes `embedded_optimizers`)
        embedded_optimizer = kwargs.pop("embedded_optimizers", embedded_optimizer) and embedded_optimizer
        self.ch, self.dim_z, self.bottom_width, self.resolution = G_ch, dim_z, bottom_width, resolution
        self.kernel_size, self.attention, self.n_classes = G_kernel_size, G_attn, n_classes
        self.G_shared = G_shared
        self.shared_dim = shared_dim if shared_dim > 0 else dim_z
        self.hier, self.cross_replica, self.mybn = hier, cross_replica, mybn
        self.activation, self.init, self.G_param, self.norm_style = G_activation, G_init, G_param, norm_style
        self.BN_eps, self.SN_eps, self.fp16 = BN_eps, SN_eps, G_fp16
        self.G_shared_feat, self.shared_dim_feat = G_shared_feat, shared_dim_feat
        self.class_cond, self.instance_cond, self.sync_bn = class_cond, instance_cond, sync_bn
        self.arch = G_arch(self.ch, self.attention)[resolution]
        n_blocks = len(self.arch["out_channels"])

        if self.hier:                       # dim_z is re-derived from the chunk size
            self.num_slots = n_blocks + 1
            self.z_chunk_size = self.dim_z // self.num_slots
            self.dim_z = self.z_chunk_size * self.num_slots
        else:
            self.num_slots, self.z_chunk_size = 1, 0

        sn_kw = dict(num_svs=num_G_SVs, num_itrs=num_G_SV_itrs, eps=self.SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn_kw)
        self.which_linear = functools.partial(layers.SNLinear, **sn_kw)
        self.which_embedding = nn.Embedding   # G's class embedding is not spectrally normalised (BigGAN.py:205-207)
        bn_linear = functools.partial(self.which_linear, bias=False) if self.G_shared else self.which_embedding
        cond_width = self.z_chunk_size if (class_cond or instance_cond) else self.n_classes
        if class_cond:
            cond_width += self.shared_dim
        if instance_cond:
            cond_width += self.shared_dim_feat
        self.which_bn = functools.partial(layers.ccbn, which_linear=bn_linear, cross_replica=self.cross_replica,
                                          mybn=self.mybn, input_size=cond_width, norm_style=self.norm_style,
                                          eps=self.BN_eps, sync_bn=sync_bn)

        self.shared = self.which_embedding(n_classes, self.shared_dim) if G_shared else layers.identity()
        self.shared_feat = self.which_linear(2048, self.shared_dim_feat) if G_shared_feat else layers.identity()
        self.linear = self.which_linear(self.dim_z // self.num_slots,
                                        self.arch["in_channels"][0] * (self.bottom_width ** 2))
        stages = []
        for i in range(n_blocks):
            stage = [layers.GBlock(in_channels=self.arch["in_channels"][i], out_channels=self.arch["out_channels"][i],
                                   which_conv=self.which_conv, which_bn=self.which_bn, activation=self.activation,
                                   upsample=(functools.partial(F.interpolate, scale_factor=2)
                                             if self.arch["upsample"][i] else None))]
            if self.arch["attention"][self.arch["resolution"][i]]:
                print("Adding attention layer in G at resolution %d" % self.arch["resolution"][i])
                stage.append(layers.Attention(self.arch["out_channels"][i], self.which_conv))
            stages.append(nn.ModuleList(stage))
        self.blocks = nn.ModuleList(stages)
        self.output_layer = _OutputLayer(
            layers.bn(self.arch["out_channels"][-1], cross_replica=self.cross_replica, mybn=self.mybn, sync_bn=sync_bn),
            self.activation, self.which_conv(self.arch["out_channels"][-1], 3))

        if not skip_init:
            self.init_weights()
        if no_optim or not embedded_optimizer:
            return
        self.lr, self.B1, self.B2, self.adam_eps = G_lr, G_B1, G_B2, adam_eps
        self.optim = FusedAdam(params=self.parameters(), lr=self.lr, betas=(self.B1, self.B2), weight_decay=0,
                               eps=self.adam_eps)

    def init_weights(self):
        self.param_count = _init_module_weights(self, self.init)
        print("Param count for G" "s initialized parameters: %d" % self.param_count)

    def get_condition_embeddings(self, cl=None, feat=None):
        """BigGAN.py:350-358."""
        parts = []
        if cl is not None:
            parts.append(self.shared(cl))
        if feat is not None:
            parts.append(self.shared_feat(feat))
        return torch.cat(parts, dim=-1) if parts else parts

    def _sn_layers(self, with_feats):
        """the spectrally normalised layers this forward is going to call"""
        skip = set() if with_feats else {id(m) for m in self.shared_feat.modules()}
        return [m for m in self.modules() if isinstance(m, layers.SN) and id(m) not in skip]

    def forward(self, z, label=None, feats=None):
        """z [B,dim_z], label [B] int64 or None, feats [B,2048] or None -> images [B,3,R,R] in [-1,1]
        (BigGAN.py:364-386)."""
        sn_layers = self._sn_layers(feats is not None)
        layers.sn_prefetch(sn_layers)
        try:
            return self._forward(z, label, feats)
        finally:
            layers.sn_drop_prefetched(sn_layers)      # no-op after a complete forward; un-bricks the module after an abort

    def _forward(self, z, label, feats):
        y = self.get_condition_embeddings(label, feats)
        if self.hier:
            zs = torch.split(z, self.z_chunk_size, 1)
            z = zs[0]
            ys = [torch.cat([y, chunk], 1) for chunk in zs[1:]]
        else:
            ys = [y] * len(self.blocks)
        h = self.linear(z)
        h = h.view(h.size(0), -1, self.bottom_width, self.bottom_width)
        for stage, y_i in zip(self.blocks, ys):
            for block in stage:
                h = block(h, y_i)
        return ops.TanhFn.apply(self.output_layer(h))


class Discriminator(nn.Module):
    def __init__(self, D_ch=64, D_wide=True, resolution=128, D_kernel_size=3, D_attn="64", n_classes=1000,
                 num_D_SVs=1, num_D_SV_itrs=1, D_activation=nn.ReLU(inplace=False), D_lr=2e-4, D_B1=0.0, D_B2=0.999,
                 adam_eps=1e-8, SN_eps=1e-12, output_dim=1, D_mixed_precision=False, D_fp16=False, D_init="ortho",
                 skip_init=False, D_param="SN", class_cond=True, embedded_optimizer=True, instance_cond=False,
                 instance_sz=2048, **kwargs):
        super().__init__()
        if D_param != "SN":
            raise NotImplementedError("ic_gan_amd.Discriminator: D_param='SN' only")
        if D_fp16 or D_mixed_precision:
            raise NotImplementedError("ic_gan_amd computes in fp32; fp16 modes are not implemented")
        embedded_optimizer = kwargs.pop("embedded_optimizers", embedded_optimizer) and embedded_optimizer
        self.ch, self.D_wide, self.resolution, self.kernel_size = D_ch, D_wide, resolution, D_kernel_size
        self.attention, self.n_classes, self.activation = D_attn, n_classes, D_activation
        self.init, self.D_param, self.SN_eps, self.fp16 = D_init, D_param, SN_eps, D_fp16
        self.arch = D_arch(self.ch, self.attention)[resolution]

        sn_kw = dict(num_svs=num_D_SVs, num_itrs=num_D_SV_itrs, eps=self.SN_eps)
        self.which_conv = functools.partial(layers.SNConv2d, kernel_size=3, padding=1, **sn_kw)
        self.which_linear = functools.partial(layers.SNLinear, **sn_kw)
        self.which_embedding = functools.partial(layers.SNEmbedding, **sn_kw)

        stages = []
        for i in range(len(self.arch["out_channels"])):
            stage = [layers.DBlock(in_channels=self.arch["in_channels"][i], out_channels=self.arch["out_channels"][i],
                                   which_conv=self.which_conv, wide=self.D_wide, activation=self.activation,
                                   preactivation=(i > 0),
                                   downsample=(nn.AvgPool2d(2) if self.arch["downsample"][i] else None))]
            if self.arch["attention"][self.arch["resolution"][i]]:
                print("Adding attention layer in D at resolution %d" % self.arch["resolution"][i])
                stage.append(layers.Attention(self.arch["out_channels"][i], self.which_conv))
            stages.append(nn.ModuleList(stage))
        self.blocks = nn.ModuleList(stages)
        top = self.arch["out_channels"][-1]
        self.linear = self.which_linear(top, output_dim)
        if class_cond and instance_cond:       # projection on class embedding ++ instance features (BigGAN.py:546-553)
            self.linear_feat = self.which_linear(instance_sz, top // 2)
            self.embed = self.which_embedding(self.n_classes, top // 2)
        elif class_cond:
            self.embed = self.which_embedding(self.n_classes, top)
        elif instance_cond:
            self.linear_feat = self.which_linear(instance_sz, top)

        if not skip_init:
            self.init_weights()
        if embedded_optimizer:
            self.lr, self.B1, self.B2, self.adam_eps = D_lr, D_B1, D_B2, adam_eps
            self.optim = FusedAdam(params=self.parameters(), lr=self.lr, betas=(self.B1, self.B2), weight_decay=0,
                                   eps=self.adam_eps)

    def init_weights(self):
        self.param_count = _init_module_weights(self, self.init)
        print("Param count for D" "s initialized parameters: %d" % self.param_count)

    def forward(self, x, y=None, feat=None):
        """x [N,3,R,R], y [N] int64 or None, feat [N,2048] or None -> logits [N,1]  (BigGAN.py:617-642)."""
        skip = set()
        if y is None and hasattr(self, "embed"):
            skip |= {id(m) for m in self.embed.modules()}
        if feat is None and hasattr(self, "linear_feat"):
            skip |= {id(m) for m in self.linear_feat.modules()}
        sn_layers = [m for m in self.modules() if isinstance(m, layers.SN) and id(m) not in skip]
        layers.sn_prefetch(sn_layers)
        try:
            return self._forward(x, y, feat)
        finally:
            layers.sn_drop_prefetched(sn_layers)

    def _forward(self, x, y, feat):
        h = x
        for stage in self.blocks:
            for block in stage:
                h = block(h)
        h = ops.ReluSumPoolFn.apply(h)                   # sum(relu(h), [2,3])
        out = self.linear(h)
        if y is not None and feat is not None:
            proj = torch.cat([self.embed(y), self.linear_feat(feat)], dim=-1)
        elif y is not None:
            proj = self.embed(y)
        elif feat is not None:
            proj = self.linear_feat(feat)
        else:
            return out
        return out + torch.sum(proj * h, 1, keepdim=True)


class G_D(nn.Module):
    """G followed by D on fake (++ real) in one module (BigGAN.py:647-711)."""

    def __init__(self, G, D, optimizer_G=None, optimizer_D=None):
        super().__init__()
        self.G, self.D = G, D
        self.optimizer_G, self.optimizer_D = optimizer_G, optimizer_D

    def generate(self, z, gy, feats_g=None):
        """the generator call of `forward` (also what train_fns.PREFETCH_NEXT_STEP issues ahead, under no_grad)"""
        return self.G(z, gy, feats_g)

    def forward(self, z, gy, feats_g=None, x=None, dy=None, feats=None, train_G=False, return_G_z=False,
                split_D=False, policy=False, DA=False, G_z=None):
        """`G_z` (not in the reference's signature): the generator's output for exactly these (z, gy, feats_g), computed ahead by the
        caller under torch.no_grad() (train_fns.PREFETCH_NEXT_STEP); only valid with train_G=False"""
        if DA:
            raise NotImplementedError("DiffAugment is disabled in every shipped IC-GAN config (SURVEY §2.1)")
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
        if x is not None:
            # both halves channels-last so the batch concatenation stays in the kernels' layout
            D_input = torch.cat([G_z, x.contiguous(memory_format=torch.channels_last)], 0)
        else:
            D_input = G_z
        D_class = torch.cat([gy, dy], 0) if dy is not None else gy
        if feats_g is not None:
            D_feats = torch.cat([feats_g, feats], 0) if feats is not None else feats_g
        else:
            D_feats = None
        D_out = self.D(D_input, D_class, D_feats)
        if x is not None:
            return torch.split(D_out, [G_z.shape[0], x.shape[0]])
        return (D_out, G_z) if return_G_z else D_out

"""IC-GAN StyleGAN2 generator / discriminator on the HIP operators  (SURVEY §8f N1).

Module classes, constructor arguments, attribute names and parameter/buffer names follow
stylegan2_ada_pytorch/training/networks.py (IC-GAN's variant: the mapping networks also embed the instance feature
vector `h`), so `state_dict()` keys match the reference's pickled networks one-to-one:

  FullyConnectedLayer 121-165   Conv2dLayer 171-242      MappingNetwork 248-355   SynthesisLayer 361-444
  ToRGBLayer 450-486            SynthesisBlock 492-645   SynthesisNetwork 651-712  Generator 718-760
  DiscriminatorBlock 766-895    MinibatchStdLayer 901-929  DiscriminatorEpilogue 935-1012  Discriminator 1018-1104

What runs where: convolutions (plain, strided, transposed, modulated) -> icg_conv2d_g_fprop / icg_conv2d_g_wgrad;
FIR resampling -> icg_upfirdn2d; bias + activation + clamp -> icg_bias_act; the small dense layers of the mapping /
affine / epilogue heads -> icg_gemm_batched (`conv2d_gradfix.linear_nt`; no vendor library call).  All operators have arbitrary-order
gradients (R1 and path-length regularisation differentiate twice).  `num_fp16_res` / `conv_clamp` (the reference's
`cfg=auto` uses 4 / 256, train.py:297-310): the highest-resolution blocks keep their activations in fp16 exactly where the
reference does (networks.py:505-515, 581-600, 793-870) -- fp16 storage through bias_act / upfirdn2d / the modulation glue,
fp16 weights, and convolutions on the fp16-input MFMA kernels (csrc/hconv.hip, csrc/hwgrad.hip: exact fp16 products, fp32
accumulation, one rounding per convolution output -- the reference's cuDNN arithmetic there); layers those kernels do not serve
(3-channel toRGB / fromRGB) run on the exact-fp32 kernels between two casts.
Fused: in the phases that differentiate once (Gmain, Dmain -- every iteration) and without autograd every
SynthesisLayer / ToRGBLayer / Conv2dLayer / FullyConnectedLayer is ONE autograd node (stylegan_ops/fused_layers.py): weights prepared
once per optimiser step, the style scale applied to the convolution's A fragments, demodulation x noise + bias + lrelu + clamp on its
accumulators (csrc/hconv.hip MOD / EP, csrc/sg2_fused.hip).  The lazy regularisers (R1, path length) differentiate twice and run the
composed operators below (stylegan_ops/modconv.py), to which every fused layer is held equal in tests/test_sg2_fused_{cpu,gpu}.py.
"""
import contextlib

import numpy as np
import torch

from ..stylegan_ops import bias_act, conv2d_gradfix, conv2d_resample, fused_layers, modulated_conv2d, upfirdn2d
from ..stylegan_ops import modconv as _modconv

_DEF_GAIN = {name: spec[2] for name, spec in bias_act.activation_funcs.items()}


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    return x * (x.square().mean(dim=dim, keepdim=True) + eps).rsqrt()


linear_nt = conv2d_gradfix.linear_nt      # x [M][K] @ w [N][K]^T on the HIP GEMM, gradients of every order


class FullyConnectedLayer(torch.nn.Module):
    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1, bias_init=0):
        super().__init__()
        self.activation = activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], np.float32(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / np.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        b = self.bias
        if fused_layers.fc_applies(x, self.weight, self.activation):       # one autograd node: GEMM, bias + activation
            return fused_layers.fc_layer(x, self.weight, b, self.activation, self.weight_gain, self.bias_gain, _DEF_GAIN[self.activation])
        if b is not None and self.bias_gain != 1:
            b = b * self.bias_gain
        # x @ w^T on the hand-written GEMM (conv2d_gradfix._Matmul: closed under differentiation, so the layer has gradients of
        # every order like the rest of the network -- path-length regularisation differentiates the affine layers twice);
        # the reference's addmm / matmul (networks.py:99-107) are cuBLAS calls
        y = linear_nt(x, self.weight, alpha=self.weight_gain)            # (w * gain) folded into the GEMM's alpha
        if self.activation == "linear" and b is not None:
            return y + b.unsqueeze(0)
        return bias_act.bias_act(y, b, act=self.activation)


class Conv2dLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation="linear", up=1, down=1,
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.activation, self.up, self.down, self.conv_clamp = activation, up, down, conv_clamp
        self.register_buffer("resample_filter", upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))
        self.act_gain = _DEF_GAIN[activation]
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size])
        bias = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(bias) if bias is not None else None
        else:
            self.register_buffer("weight", weight)
            if bias is not None:
                self.register_buffer("bias", bias)
            else:
                self.bias = None

    def forward(self, x, gain=1):
        pl = fused_layers.conv_applies(x, self.weight, self.activation, self.up, self.down, self.padding,
                                       int(self.resample_filter.shape[-1]), self.up == 1)
        if pl is not None:      # one autograd node: [FIR] convolution [FIR], bias + activation + clamp; weights prepared once per step
            return fused_layers.conv_layer(self, x, self.weight, self.bias, self.resample_filter, pl, self.activation, self.weight_gain,
                                           self.act_gain * gain, self.conv_clamp * gain if self.conv_clamp is not None else None)
        x = conv2d_resample.conv2d_resample(x=x, w=(self.weight * self.weight_gain).to(x.dtype), f=self.resample_filter,
                                            up=self.up, down=self.down, padding=self.padding, flip_weight=(self.up == 1))
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, self.bias, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim, c_dim, h_dim, w_dim, num_ws, num_layers=8, embed_features=None, embed_features_feat=None,
                 layer_features=None, activation="lrelu", lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.z_dim, self.c_dim, self.h_dim, self.w_dim = z_dim, c_dim, h_dim, w_dim
        self.num_ws, self.num_layers, self.w_avg_beta = num_ws, num_layers, w_avg_beta
        embed_features = 0 if c_dim == 0 else (w_dim if embed_features is None else embed_features)
        embed_features_feat = 0 if h_dim == 0 else (w_dim if embed_features_feat is None else embed_features_feat)
        layer_features = w_dim if layer_features is None else layer_features
        widths = [z_dim + embed_features + embed_features_feat] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        if h_dim > 0:
            self.embed_feats = FullyConnectedLayer(h_dim, embed_features_feat)
        for idx in range(num_layers):
            setattr(self, f"fc{idx}", FullyConnectedLayer(widths[idx], widths[idx + 1], activation=activation,
                                                          lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer("w_avg", torch.zeros([w_dim]))

    def forward(self, z, c, h, truncation_psi=1, truncation_cutoff=None, skip_w_avg_update=False):
        parts = []
        if self.z_dim > 0:
            assert z.shape[1] == self.z_dim
            parts.append(normalize_2nd_moment(z.to(torch.float32)))
        cond = []
        if self.c_dim > 0:
            assert c.shape[1] == self.c_dim
            cond.append(self.embed(c.to(torch.float32)))
        if self.h_dim > 0:
            assert h.shape[1] == self.h_dim
            cond.append(self.embed_feats(h.to(torch.float32)))
        if cond:      # class and instance embeddings are normalised jointly (networks.py:304-313)
            parts.append(normalize_2nd_moment(torch.cat(cond, dim=1) if len(cond) > 1 else cond[0]))
        x = torch.cat(parts, dim=1) if len(parts) > 1 else parts[0]
        for idx in range(self.num_layers):
            x = getattr(self, f"fc{idx}")(x)
        if self.w_avg_beta is not None and self.training and not skip_w_avg_update:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            assert self.w_avg_beta is not None
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True,
                 activation="lrelu", resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        self.resolution, self.up, self.use_noise = resolution, up, use_noise
        self.activation, self.conv_clamp = activation, conv_clamp
        self.register_buffer("resample_filter", upfirdn2d.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = _DEF_GAIN[activation]
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        if use_noise:
            self.register_buffer("noise_const", torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode="random", fused_modconv=True, gain=1):
        assert noise_mode in ["random", "const", "none"]
        assert x.shape[1] == self.weight.shape[1] and x.shape[2] == self.resolution // self.up
        pl = None
        if not (fused_modconv and _modconv.GROUPED_FUSED_MODCONV):      # (the literal grouped route is a request for that op graph)
            pl = fused_layers.modconv_applies(x, self.weight, w, self.affine.weight, self.up, self.padding,
                                              int(self.resample_filter.shape[-1]), self.up == 1)
        if pl is not None:      # one autograd node for the whole layer (stylegan_ops/fused_layers.py)
            base, bstride = None, 0
            if self.use_noise and noise_mode == "random":
                base, bstride = _randn([x.shape[0], 1, self.resolution, self.resolution], x.device), self.resolution ** 2
            if self.use_noise and noise_mode == "const":
                base = self.noise_const
            return fused_layers.modconv_layer(self, x, w, self.affine, self.weight, self.noise_strength if self.use_noise else None,
                                              self.bias, base, bstride, self.resample_filter, pl, self.act_gain * gain,
                                              self.conv_clamp * gain if self.conv_clamp is not None else None)
        styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == "random":
            noise = _randn([x.shape[0], 1, self.resolution, self.resolution], x.device) * self.noise_strength
        if self.use_noise and noise_mode == "const":
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, noise=noise, up=self.up, padding=self.padding,
                             resample_filter=self.resample_filter, flip_weight=(self.up == 1),
                             fused_modconv=fused_modconv)
        clamp = self.conv_clamp * gain if self.conv_clamp is not None else None
        return bias_act.bias_act(x, self.bias, act=self.activation, gain=self.act_gain * gain, clamp=clamp)


def _randn(shape, device):
    """single entry point for in-network random numbers (tests substitute a host-seeded generator)."""
    return torch.randn(shape, device=device)


class ToRGBLayer(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.conv_clamp = conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / np.sqrt(in_channels * (kernel_size ** 2))

    def forward(self, x, w, fused_modconv=True, img=None):
        """`img` (an extension of the reference signature, networks.py:476): the fp32 image to accumulate into; returns img + y"""
        if fused_layers.torgb_applies(x, self.weight, w):      # one pass over x, image accumulation included
            return fused_layers.torgb_layer(x, w, self.affine, self.weight, self.bias, img, self.weight_gain, self.conv_clamp)
        styles = self.affine(w) * self.weight_gain
        x = modulated_conv2d(x=x, weight=self.weight, styles=styles, demodulate=False, fused_modconv=fused_modconv)
        y = bias_act.bias_act(x, self.bias, clamp=self.conv_clamp)
        if img is None:
            return y
        return img + y.to(torch.float32)


class SynthesisBlock(torch.nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture="skip",
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, use_fp16=False, fp16_channels_last=False,
                 **layer_kwargs):
        assert architecture in ["orig", "skip", "resnet"]
        super().__init__()
        self.in_channels, self.w_dim, self.resolution = in_channels, w_dim, resolution
        self.img_channels, self.is_last, self.architecture = img_channels, is_last, architecture
        self.use_fp16 = use_fp16
        self.register_buffer("resample_filter", upfirdn2d.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2,
                                        resample_filter=resample_filter, conv_clamp=conv_clamp, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution,
                                    conv_clamp=conv_clamp, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == "skip":
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp)
            self.num_torgb += 1
        if in_channels != 0 and architecture == "resnet":
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2,
                                    resample_filter=resample_filter)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, **layer_kwargs):
        assert ws.shape[1] == self.num_conv + self.num_torgb and ws.shape[2] == self.w_dim
        w_iter = iter(ws.unbind(dim=1))
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32      # networks.py:581
        if fused_modconv is None:
            fused_modconv = not self.training
        if self.in_channels == 0:
            x = self.const.to(dtype).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
        else:
            x = x.to(dtype)
        if self.in_channels == 0:
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        elif self.architecture == "resnet":
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, gain=np.sqrt(0.5), **layer_kwargs)
            x = y + x
        else:
            x = self.conv0(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
            x = self.conv1(x, next(w_iter), fused_modconv=fused_modconv, **layer_kwargs)
        if img is not None:
            img = upfirdn2d.upsample2d(img, self.resample_filter)
        if self.is_last or self.architecture == "skip":
            # the image accumulates in fp32 (networks.py:630); the addition rides in the toRGB layer
            img = self.torgb(x, next(w_iter), fused_modconv=fused_modconv, img=img).to(torch.float32)
        assert x.dtype == dtype and (img is None or img.dtype == torch.float32)
        return x, img


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=0,
                 **block_kwargs):
        assert img_resolution >= 4 and img_resolution & (img_resolution - 1) == 0
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)          # networks.py:665
        self.num_ws = 0
        for res in self.block_resolutions:
            block = SynthesisBlock(channels[res // 2] if res > 4 else 0, channels[res], w_dim=w_dim, resolution=res,
                                   img_channels=img_channels, is_last=(res == img_resolution),
                                   use_fp16=(res >= fp16_resolution), **block_kwargs)
            self.num_ws += block.num_conv + (block.num_torgb if res == img_resolution else 0)
            setattr(self, f"b{res}", block)

    def forward(self, ws, **block_kwargs):
        assert ws.shape[1] == self.num_ws and ws.shape[2] == self.w_dim
        ws = ws.to(torch.float32)
        x = img = None
        w_idx = 0
        for res in self.block_resolutions:
            block = getattr(self, f"b{res}")
            x, img = block(x, img, ws.narrow(1, w_idx, block.num_conv + block.num_torgb), **block_kwargs)
            w_idx += block.num_conv
        return img


class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, h_dim, w_dim, img_resolution, img_channels, mapping_kwargs={},
                 synthesis_kwargs={}):
        super().__init__()
        self.z_dim, self.c_dim, self.h_dim, self.w_dim = z_dim, c_dim, h_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                                          **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, h_dim=h_dim, w_dim=w_dim, num_ws=self.num_ws,
                                      **mapping_kwargs)

    def forward(self, z, c, feats, truncation_psi=1, truncation_cutoff=None, **synthesis_kwargs):
        ws = self.mapping(z, c, feats, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, **synthesis_kwargs)


class DiscriminatorBlock(torch.nn.Module):
    def __init__(self, in_channels, tmp_channels, out_channels, resolution, img_channels, first_layer_idx,
                 architecture="resnet", activation="lrelu", resample_filter=[1, 3, 3, 1], conv_clamp=None,
                 use_fp16=False, fp16_channels_last=False, freeze_layers=0):
        assert in_channels in [0, tmp_channels]
        assert architecture in ["orig", "skip", "resnet"]
        super().__init__()
        self.in_channels, self.resolution, self.img_channels = in_channels, resolution, img_channels
        self.first_layer_idx, self.architecture, self.use_fp16 = first_layer_idx, architecture, use_fp16
        self.register_buffer("resample_filter", upfirdn2d.setup_filter(resample_filter))
        self.num_layers = 0

        def trainable():
            layer_idx = self.first_layer_idx + self.num_layers
            self.num_layers += 1
            return layer_idx >= freeze_layers

        if in_channels == 0 or architecture == "skip":
            self.fromrgb = Conv2dLayer(img_channels, tmp_channels, kernel_size=1, activation=activation,
                                       trainable=trainable(), conv_clamp=conv_clamp)
        self.conv0 = Conv2dLayer(tmp_channels, tmp_channels, kernel_size=3, activation=activation,
                                 trainable=trainable(), conv_clamp=conv_clamp)
        self.conv1 = Conv2dLayer(tmp_channels, out_channels, kernel_size=3, activation=activation, down=2,
                                 trainable=trainable(), resample_filter=resample_filter, conv_clamp=conv_clamp)
        if architecture == "resnet":
            self.skip = Conv2dLayer(tmp_channels, out_channels, kernel_size=1, bias=False, down=2,
                                    trainable=trainable(), resample_filter=resample_filter)

    def forward(self, x, img, force_fp32=False):
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32      # networks.py:854
        if x is not None:
            assert x.shape[1] == self.in_channels and x.shape[2] == self.resolution
            x = x.to(dtype)
        if self.in_channels == 0 or self.architecture == "skip":
            assert img.shape[1] == self.img_channels and img.shape[2] == self.resolution
            img = img.to(dtype)
            y = self.fromrgb(img)
            x = x + y if x is not None else y
            img = upfirdn2d.downsample2d(img, self.resample_filter) if self.architecture == "skip" else None
        if self.architecture == "resnet":
            y = self.skip(x, gain=np.sqrt(0.5))
            x = self.conv0(x)
            x = self.conv1(x, gain=np.sqrt(0.5))
            x = y + x
        else:
            x = self.conv0(x)
            x = self.conv1(x)
        assert x.dtype == dtype
        return x, img


_SEPARATE_BATCHES = 1


@contextlib.contextmanager
def separate_batches(n):
    """inside: the batch a Discriminator sees is n independent batches laid end to end (loss.py runs D on [generated; real] in one
    pass); the only layer that looks across samples -- MinibatchStdLayer -- then forms its groups inside each part"""
    global _SEPARATE_BATCHES
    old, _SEPARATE_BATCHES = _SEPARATE_BATCHES, int(n)
    try:
        yield
    finally:
        _SEPARATE_BATCHES = old


class MinibatchStdLayer(torch.nn.Module):
    def __init__(self, group_size, num_channels=1):
        super().__init__()
        self.group_size, self.num_channels = group_size, num_channels

    def forward(self, x):
        if _SEPARATE_BATCHES > 1:
            # (no silent fallback to joint statistics: groups mixing generated and real samples would give other gradients than the
            # reference's two passes)
            assert x.shape[0] % _SEPARATE_BATCHES == 0, "separate_batches(%d): batch of %d does not split" % (_SEPARATE_BATCHES, x.shape[0])
            return torch.cat([self._one(part) for part in x.chunk(_SEPARATE_BATCHES)])
        return self._one(x)

    def _one(self, x):
        N, C, H, W = x.shape
        G = min(self.group_size, N) if self.group_size is not None else N
        F = self.num_channels
        c = C // F
        y = x.reshape(G, -1, F, c, H, W)           # groups of G samples, F statistics over c channels each
        y = y - y.mean(dim=0)
        y = (y.square().mean(dim=0) + 1e-8).sqrt()
        y = y.mean(dim=[2, 3, 4]).reshape(-1, F, 1, 1).repeat(G, 1, H, W)
        return torch.cat([x, y], dim=1)


class DiscriminatorEpilogue(torch.nn.Module):
    def __init__(self, in_channels, cmap_dim, resolution, img_channels, architecture="resnet", mbstd_group_size=4,
                 mbstd_num_channels=1, activation="lrelu", conv_clamp=None):
        assert architecture in ["orig", "skip", "resnet"]
        super().__init__()
        self.in_channels, self.cmap_dim, self.resolution = in_channels, cmap_dim, resolution
        self.img_channels, self.architecture = img_channels, architecture
        if architecture == "skip":
            self.fromrgb = Conv2dLayer(img_channels, in_channels, kernel_size=1, activation=activation)
        self.mbstd = MinibatchStdLayer(mbstd_group_size, mbstd_num_channels) if mbstd_num_channels > 0 else None
        self.conv = Conv2dLayer(in_channels + mbstd_num_channels, in_channels, kernel_size=3, activation=activation,
                                conv_clamp=conv_clamp)
        self.fc = FullyConnectedLayer(in_channels * (resolution ** 2), in_channels, activation=activation)
        self.out = FullyConnectedLayer(in_channels, 1 if cmap_dim == 0 else cmap_dim)

    def forward(self, x, img, cmap, force_fp32=False):
        assert x.shape[1] == self.in_channels and x.shape[2] == self.resolution
        x = x.to(torch.float32)                      # the epilogue always runs in fp32 (networks.py:982-985)
        if self.architecture == "skip":
            x = x + self.fromrgb(img.to(torch.float32))
        if self.mbstd is not None:
            x = self.mbstd(x)
        x = self.conv(x)
        x = self.fc(x.flatten(1))       # NCHW flattening order, like the reference (x is logically NCHW)
        x = self.out(x)
        if self.cmap_dim > 0:
            assert cmap.shape[1] == self.cmap_dim
            x = (x * cmap).sum(dim=1, keepdim=True) * (1 / np.sqrt(self.cmap_dim))
        return x


class Discriminator(torch.nn.Module):
    def __init__(self, c_dim, h_dim, img_resolution, img_channels, architecture="resnet", channel_base=32768,
                 channel_max=512, num_fp16_res=0, conv_clamp=None, cmap_dim=None, block_kwargs={}, mapping_kwargs={},
                 epilogue_kwargs={}):
        super().__init__()
        self.c_dim, self.h_dim, self.img_resolution, self.img_channels = c_dim, h_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(self.img_resolution_log2, 2, -1)]
        channels = {res: min(channel_base // res, channel_max) for res in self.block_resolutions + [4]}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)          # networks.py:1045
        if cmap_dim is None:
            cmap_dim = channels[4]
        if c_dim == 0 and h_dim == 0:
            cmap_dim = 0
        common = dict(img_channels=img_channels, architecture=architecture, conv_clamp=conv_clamp)
        cur_layer_idx = 0
        for res in self.block_resolutions:
            block = DiscriminatorBlock(channels[res] if res < img_resolution else 0, channels[res], channels[res // 2],
                                       resolution=res, first_layer_idx=cur_layer_idx, use_fp16=(res >= fp16_resolution),
                                       **block_kwargs, **common)
            setattr(self, f"b{res}", block)
            cur_layer_idx += block.num_layers
        if c_dim > 0 or h_dim > 0:
            self.mapping = MappingNetwork(z_dim=0, c_dim=c_dim, h_dim=h_dim, w_dim=cmap_dim, num_ws=None,
                                          w_avg_beta=None, **mapping_kwargs)
        self.b4 = DiscriminatorEpilogue(channels[4], cmap_dim=cmap_dim, resolution=4, **epilogue_kwargs, **common)

    def forward(self, img, c, h, **block_kwargs):
        x = None
        for res in self.block_resolutions:
            x, img = getattr(self, f"b{res}")(x, img, **block_kwargs)
        cmap = self.mapping(None, c, h) if (self.c_dim > 0 or self.h_dim > 0) else None
        return self.b4(x, img, cmap)

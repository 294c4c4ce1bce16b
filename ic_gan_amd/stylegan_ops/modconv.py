"""modulated_conv2d  (stylegan2_ada_pytorch/training/networks.py:37-117).

Style modulation, convolution (with optional resampling), demodulation and noise.  Executed as "scale the activations
before and after the convolution" (the reference's fused_modconv=False branch, networks.py:78-97, which is what training
uses); `fused_modconv=True` — a grouped convolution over per-sample weights in the reference, used in eval mode — is
computed through the same branch by default: x*s -> conv(w) -> *d equals conv(w*s*d) exactly in real arithmetic and to fp32
rounding here, and needs no per-sample weight tensor [N, O, I, k, k] in HBM.  `GROUPED_FUSED_MODCONV = True` switches a
`fused_modconv=True` call to the reference's literal formulation (networks.py:64-73,100-117: per-sample weights, one group per
sample through `conv2d_resample(groups=N)`) — N convolutions of batch 1 instead of one of batch N, there for parity checks
against the reference's fused goldens and for callers who ask for exactly that op graph.

Role since round 5: this composed form -- closed under differentiation -- serves the phases that differentiate TWICE (path-length and R1
regularisation) and every shape the fused layers do not take; the first-order phases and inference run a SynthesisLayer / ToRGBLayer
as one autograd node with modulation and demodulation inside the convolution kernel (fused_layers.py), which the tests hold equal to
this function (outputs and all first-order gradients, fp32 and fp16)."""
import numpy as np
import torch

from . import conv2d_gradfix, conv2d_resample, fma

GROUPED_FUSED_MODCONV = False        # True: fused_modconv=True runs the reference's grouped convolution (see the module docstring)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=True, fused_modconv=True):
    batch_size = int(x.shape[0])
    out_channels, in_channels, kh, kw = (int(s) for s in weight.shape)
    assert x.shape[1] == in_channels and tuple(styles.shape) == (batch_size, in_channels)
    if x.dtype not in (torch.float32, torch.float16):
        raise NotImplementedError("modulated_conv2d: fp32 or fp16 activations")

    # fp16: pre-normalise weights and styles so that the modulated activations cannot overflow (networks.py:57-63)
    if x.dtype == torch.float16 and demodulate:
        weight = weight * (1 / np.sqrt(in_channels * kh * kw) / weight.norm(float("inf"), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float("inf"), dim=1, keepdim=True)

    if fused_modconv and GROUPED_FUSED_MODCONV:
        w = weight.unsqueeze(0) * styles.reshape(batch_size, 1, -1, 1, 1)                        # [N, O, I, k, k]
        if demodulate:
            w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(batch_size, -1, 1, 1, 1)
        x = x.reshape(1, -1, *x.shape[2:])                                                      # one group per sample
        x = conv2d_resample.conv2d_resample(x=x, w=w.reshape(-1, in_channels, kh, kw).to(x.dtype), f=resample_filter, up=up,
                                            down=down, padding=padding, groups=batch_size, flip_weight=flip_weight)
        x = x.reshape(batch_size, -1, *x.shape[2:])
        return x if noise is None else x + noise.to(x.dtype)

    dcoefs = None
    if demodulate:
        # d[n, o] = rsqrt(sum_{i,k} (w[o,i,k] s[n,i])^2 + 1e-8)  (networks.py:70-75) without materialising [N,O,I,k,k]
        wsq = weight.square().sum(dim=[2, 3])                              # [O, I]
        dcoefs = (conv2d_gradfix.linear_nt(styles.square().float(), wsq.float()) + 1e-8).rsqrt()    # [N, O], on the HIP GEMM

    x = x * styles.to(x.dtype).reshape(batch_size, -1, 1, 1)
    x = conv2d_resample.conv2d_resample(x=x, w=weight.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding,
                                        flip_weight=flip_weight)
    if demodulate and noise is not None:
        x = fma.fma(x, dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1), noise.to(x.dtype))
    elif demodulate:
        x = x * dcoefs.to(x.dtype).reshape(batch_size, -1, 1, 1)
    elif noise is not None:
        x = x + noise.to(x.dtype)
    return x

"""fma(a, b, c) = a * b + c with broadcasting  (stylegan2_ada_pytorch/torch_utils/ops/fma.py:17-70).

The reference's op is `torch.addcmul` wrapped in an autograd.Function whose only purpose is a leaner backward graph;
it has no native code.  Same here: one `torch.addcmul` (arbitrary-order gradients come for free).  It serves the composed
(twice-differentiable) modulated convolution of the regulariser phases; in the first-order phases and in inference its hot use --
demodulation + noise after a modulated convolution -- runs on the convolution's accumulators (csrc/hconv.hip EP, fused_layers.py)."""
import torch


def fma(a, b, c):
    return torch.addcmul(c, a, b)

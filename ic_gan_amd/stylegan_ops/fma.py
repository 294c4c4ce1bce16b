"""fma(a, b, c) = a * b + c with broadcasting  (stylegan2_ada_pytorch/torch_utils/ops/fma.py:17-70).

The reference's op is `torch.addcmul` wrapped in an autograd.Function whose only purpose is a leaner backward graph;
it has no native code.  Same here: elementwise glue stays in PyTorch (arbitrary-order gradients come for free); the
hot use — demodulation + noise after a modulated convolution — is the epilogue of the convolution it follows."""
import torch


def fma(a, b, c):
    return torch.addcmul(c, a, b)

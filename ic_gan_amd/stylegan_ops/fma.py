"""fma(a, b, c) = a * b + c with broadcasting  (stylegan2_ada_pytorch/torch_utils/ops/fma.py:17-70).

The reference's op is `torch.addcmul` wrapped in an autograd.Function whose only purpose is a leaner backward graph;
it has no native code.  Same here: one `torch.addcmul` (arbitrary-order gradients come for free).  Its hot use --
demodulation + noise after a modulated convolution -- is a separate elementwise pass, NOT an epilogue of that
convolution (SURVEY 8(f) N1 asks for the fusion; it is not built)."""
import torch


def fma(a, b, c):
    return torch.addcmul(c, a, b)

"""Shared case tables for the StyleGAN2 custom-op goldens (generator script + tests)."""
import numpy as np
import torch

ACTS = ["linear", "relu", "lrelu", "tanh", "sigmoid", "elu", "selu", "softplus", "swish"]
UPFIR = [  # N, C, H, W, filter taps (1-D -> outer product unless sep), up, down, padding, flip, gain
    (2, 3, 8, 8, [1, 3, 3, 1], 2, 1, [2, 1, 2, 1], False, 4.0),
    (2, 5, 9, 7, [1, 3, 3, 1], 1, 2, [1, 1, 1, 1], False, 1.0),
    (1, 4, 16, 16, [1, 2, 3, 4], 1, 1, [1, 2, 2, 1], True, 1.0),
    (2, 2, 6, 6, [1, 4, 6, 4, 1], 2, 2, [0, 3, -1, 2], False, 0.7),
    (1, 8, 17, 17, [1, 3, 3, 1], 1, 1, [-1, -1, 0, 0], False, 1.0),
    (2, 3, 8, 8, [1, 2, 3, 4, 4, 3, 2, 1], 2, 1, [4, 3, 4, 3], True, 2.0),      # 8 taps -> separable in the reference
]


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)



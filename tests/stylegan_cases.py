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




# ---- a19: conv2d_gradfix / conv2d_resample / modulated_conv2d ------------------------------------------------------
CONV = [  # N, Cin, H, W, Cout, R, stride, padding, transpose, output_padding
    (2, 8, 9, 9, 12, 3, 1, 1, False, 0),
    (2, 16, 8, 8, 16, 3, 2, 0, False, 0),
    (1, 4, 11, 7, 6, 1, 1, 0, False, 0),
    (2, 8, 10, 10, 4, 3, 2, 1, False, 0),        # (H + 2p - R) % stride != 0: the data gradient needs output_padding
    (2, 5, 6, 6, 3, 3, 1, 2, False, 0),          # "full" padding
    (2, 8, 5, 5, 6, 3, 2, 0, True, 0),
    (2, 16, 4, 4, 16, 3, 2, 1, True, 1),
    (1, 4, 6, 6, 4, 1, 1, 0, True, 0),
    (2, 8, 4, 4, 8, 4, 2, 1, True, 0),
    (2, 32, 16, 16, 32, 3, 1, 1, False, 0),      # Cin % 16 == 0, pow2 grid: the fast loader paths
    (2, 32, 16, 16, 32, 3, 2, 1, True, 1),
]
RESAMPLE = [  # N, Cin, H, W, Cout, k, up, down, padding, flip_weight, flip_filter, filter taps
    (2, 8, 8, 8, 6, 3, 2, 1, 1, False, False, [1, 3, 3, 1]),     # SynthesisLayer conv0 (up)
    (2, 8, 8, 8, 6, 3, 1, 2, 1, True, False, [1, 3, 3, 1]),      # DiscriminatorBlock conv1 (down)
    (2, 8, 8, 8, 6, 1, 1, 2, 0, True, False, [1, 3, 3, 1]),      # DiscriminatorBlock skip (1x1, down)
    (2, 8, 8, 8, 6, 1, 2, 1, 0, True, False, [1, 3, 3, 1]),      # 1x1 + up
    (2, 8, 8, 8, 6, 3, 1, 1, 1, True, False, None),              # plain 3x3
    (2, 8, 8, 8, 3, 1, 1, 1, 0, True, False, None),              # ToRGB
    (1, 4, 6, 6, 4, 3, 2, 2, 1, True, True, [1, 2, 3, 1]),       # up and down, asymmetric filter, flipped
    (1, 4, 7, 7, 4, 3, 1, 1, [2, 0, 1, 1], False, False, None),  # asymmetric padding -> generic fallback
]
MODCONV = [  # N, Cin, H, W, Cout, k, up, demodulate, noise, fused_modconv
    (3, 8, 8, 8, 6, 3, 1, True, True, False),
    (3, 8, 8, 8, 6, 3, 2, True, True, False),
    (2, 8, 8, 8, 3, 1, 1, False, False, False),      # ToRGB
    (2, 8, 4, 4, 8, 3, 1, True, False, True),        # eval-mode fused request
    (2, 8, 4, 4, 8, 3, 2, True, True, True),
]

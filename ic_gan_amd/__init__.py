"""ic_gan_amd — MI355X-native engine for the IC-GAN G+D training / sampling hot path.

Python host code (this package) mirrors the reference's plugin seam — ``config["model"]`` -> module with
``Generator`` / ``Discriminator`` / ``G_D`` (BigGAN_PyTorch/trainer.py:122), ``train_fns.GAN_training_function``
and the checkpoint layout — and calls hand-written gfx950 kernels through the C-ABI of
``ic_gan_amd/lib/libicgan_hip.so`` (include/icgan_hip.h).  There is no CPU or eager fallback.
"""
from . import _lib  # noqa: F401  (import does not load the .so; first kernel call does, loudly)

__all__ = ["BigGAN", "layers", "ops", "losses", "train_fns", "utils", "optim", "_lib"]

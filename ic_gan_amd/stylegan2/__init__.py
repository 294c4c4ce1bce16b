"""IC-GAN's StyleGAN2 backbone on the MI355X operators (networks, loss, training step)."""
from . import networks, loss, training_step  # noqa: F401

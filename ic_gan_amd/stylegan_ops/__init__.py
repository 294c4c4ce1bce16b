"""MI355X versions of the StyleGAN2-ADA custom ops used by IC-GAN's StyleGAN2 backbone
(stylegan2_ada_pytorch/torch_utils/ops): same Python API, backed by icg_bias_act / icg_upfirdn2d."""
from . import bias_act, upfirdn2d  # noqa: F401

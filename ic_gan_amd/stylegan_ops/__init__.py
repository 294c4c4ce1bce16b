"""MI355X versions of the StyleGAN2-ADA custom ops used by IC-GAN's StyleGAN2 backbone
(stylegan2_ada_pytorch/torch_utils/ops): same Python API, backed by icg_bias_act / icg_upfirdn2d /
icg_conv2d_g_fprop / icg_conv2d_g_wgrad."""
from . import bias_act, upfirdn2d, conv2d_gradfix, conv2d_resample, fma  # noqa: F401
from .modconv import modulated_conv2d  # noqa: F401

"""2-D convolution with optional up/down-sampling  (stylegan2_ada_pytorch/torch_utils/ops/conv2d_resample.py:79-216).

Same argument meaning and the same decomposition as the reference, because parity (including second-order gradients)
is defined on that op graph:
    1x1 + down     : upfirdn2d(down)           -> conv                     (129-135)
    1x1 + up       : conv                      -> upfirdn2d(up, gain up^2)  (138-149)
    kxk + down     : upfirdn2d(pad, blur)      -> conv(stride=down)         (152-160)
    kxk + up       : conv_transpose2d(stride=up) -> upfirdn2d(blur, gain up^2) [-> upfirdn2d(down)]   (163-197)
    no resampling  : conv(padding)                                          (200-204)
    otherwise      : upfirdn2d(up) -> conv -> upfirdn2d(down)               (207-216)
Convolutions run on icg_conv2d_g_fprop / icg_conv2d_g_wgrad (conv2d_gradfix.py here), FIR resampling on
icg_upfirdn2d.  groups > 1: one convolution per group (conv2d_gradfix._per_group)."""
import torch

from . import conv2d_gradfix, upfirdn2d
from .upfirdn2d import _get_filter_size, _parse_padding


def _conv(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d_resample.py:35-73 without the cuDNN work-arounds (which only re-route 1x1 convolutions)."""
    if not flip_weight:          # F.conv2d correlates; flip_weight=False asks for a true convolution
        w = w.flip([2, 3])
    op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    # fp16 blocks (`num_fp16_res`, training/networks.py:505-515,593-600): fp16 activations and fp16 weights go to conv2d_gradfix as
    # they are -- it contracts them on the fp16-input MFMA kernel with fp32 accumulation, the reference's arithmetic there
    return op(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    groups = int(groups)
    out_channels, in_channels, kh, kw = (int(s) for s in w.shape)          # in_channels: per group
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    if up > 1:      # padding is specified with respect to the upsampled image
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, groups=groups, flip_weight=flip_weight)

    if kw == 1 and kh == 1 and up > 1 and down == 1:
        x = _conv(x, w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv(x, w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:
        if groups == 1:
            wt = w.transpose(0, 1)
        else:     # F.conv_transpose2d's grouped layout [G * Cin_g][Cout_g][k][k]: swap the two channel axes inside every group
            wt = w.reshape(groups, out_channels // groups, in_channels, kh, kw).transpose(1, 2)
            wt = wt.reshape(groups * in_channels, out_channels // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv(x, wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                                flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv(x, w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2,
                            flip_filter=flip_filter)
    x = _conv(x, w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x

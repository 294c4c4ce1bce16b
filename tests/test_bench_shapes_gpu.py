"""Per-layer parity AT BENCH SHAPES (missing #1 / weak #2): the convolution routes that dominate the cfg3
benchmark, run on the MI355X at the benchmark's own layer shapes and batch sizes and compared with the CPU's
`F.conv2d` / `torch.nn.grad.conv2d_input` / `torch.nn.grad.conv2d_weight` (fp32, the arithmetic the reference's
`layers.SNConv2d.forward` reaches through ATen, BigGAN_PyTorch/layers.py:144-153):

  96 -> 96   @ 256x256, B = 128   plain 3x3        F(4x4,3x3), 36 planes         (D block 0 conv2 / G block 5 conv2)
  192 -> 96  @ 128 -> 256, B = 64 upsample-fused   25-plane domain               (G block 5 conv1)
  96 -> 192  @ 256 -> 128, B = 64 avgpool-fused    25-plane domain               (D block 1 conv2)
  1536->1536 @ 8x8, B = 64        plain 3x3        F(4x4,3x3), 2x2 tiles / image (G block 0 conv2)
  768 -> 384 @ 32 -> 64, B = 64   upsample-fused   25-plane domain               (G block 3 conv1)
  256 -> 32  @ 256x256, B = 128   plain 3x3        implicit GEMM with 64-bit offsets (`small == false`, gemm_conv.hip):
                                                   2^31 input elements

Forward and data gradient are linear per sample: they are checked on three batch entries (first, middle, last) of the
full-batch launch.  The weight gradient sums over the batch: checked against the CPU on the FULL batch.
Tolerances (rel. L2 of the whole tensor, written per route below): 1e-5 forward / data gradient on every route, 2e-5 weight
gradient (K = B*H*W up to 8.4M terms, fp32 on both sides); measured 3e-7 ... 3.6e-6 (profiles/r02_bench_shape_parity.txt)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
PRE_RELU = 1
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_shape_parity.txt")


def _L():
    import ic_gan_amd._lib as L
    return L


def _note(line):
    print(line, flush=True)
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def rel_l2(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).norm() / (ref.norm() + 1e-300))


def max_rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-300))


def gpu_randn_cl(b, c, h, w, seed):
    """logical NCHW, channels-last memory, generated on the device (the host never holds the full tensor)"""
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.randn(b, h, w, c, device="cuda", generator=g).permute(0, 3, 1, 2)


def _weights(cout, cin, seed):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, 3, 3, cin, generator=g) / np.sqrt(9 * cin)          # OHWI
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()                         # dgrad layout [Cin][3][3][Cout], taps flipped
    return w, wd, w.permute(0, 3, 1, 2).contiguous()                           # + OIHW for the CPU reference


def _bytes(n):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device="cuda")


def _samples(b):
    return sorted({0, b // 2, b - 1})


def _check(tag, got, ref, tol):
    r, m = rel_l2(got, ref), max_rel(got, ref)
    _note(f"BENCHSHAPE {tag:58s} rel_l2 {r:.3e}  max|err|/max|ref| {m:.3e}  (bound {tol:.0e})")
    assert np.isfinite(r) and r < tol, (tag, r, tol)


# ------------------------------------------------------------------------------------------------ plain 3x3, F(4x4,3x3)
@pytest.mark.parametrize("B,H,Cin,Cout", [(128, 256, 96, 96), (64, 8, 1536, 1536)])
def test_plain_winograd4_at_bench_shape(B, H, Cin, Cout):
    L = _L()
    import ic_gan_amd.ops as ops
    assert ops.winograd_applies(Cin, Cout, H, H, B) == 4, "production routing changed: this shape must take F(4x4,3x3)"
    w, wd, w_oihw = _weights(Cout, Cin, 1)
    x = gpu_randn_cl(B, Cin, H, H, 2)
    dy = gpu_randn_cl(B, Cout, H, H, 3)
    U, Ud = torch.empty(36 * Cout * Cin, device="cuda"), torch.empty(36 * Cout * Cin, device="cuda")
    L.call("icg_wino4_weight_transform", w.cuda(), U, Cout, Cin)
    L.call("icg_wino4_weight_transform", wd.cuda(), Ud, Cin, Cout)
    bias = torch.linspace(-1, 1, Cout)
    # forward (ReLU prologue, bias)
    nb = L.query("icg_conv2d_wino4_workspace_bytes", B, H, H, Cin, Cout)
    ws = _bytes(nb)
    out = torch.empty_like(dy)
    L.call("icg_conv2d_wino4_fprop", x, U, bias.cuda(), None, out, None, None, 0, B, H, H, Cin, Cout, PRE_RELU | L.ICG_WINO_KEEP_V, 1.0, ws, nb)   # V is read back below
    v_saved = ws[: 36 * B * (H // 4) * (H // 4) * Cin * 4].view(torch.float32)
    # data gradient = the same entry on dy with the dgrad-layout weights
    nbd = L.query("icg_conv2d_wino4_workspace_bytes", B, H, H, Cout, Cin)
    da = torch.empty_like(x)
    L.call("icg_conv2d_wino4_fprop", dy, Ud, None, None, da, None, None, 0, B, H, H, Cout, Cin, 0, 1.0, _bytes(nbd), nbd)
    for i in _samples(B):
        xi = F.relu(x[i:i + 1].cpu().contiguous())
        _check(f"wino4 fprop {Cin}->{Cout}@{H} B{B} sample {i}", out[i:i + 1], F.conv2d(xi, w_oihw, bias, padding=1), 1e-5)
        ref = torch.nn.grad.conv2d_input((1, Cin, H, H), w_oihw, dy[i:i + 1].cpu().contiguous(), padding=1)
        _check(f"wino4 dgrad {Cin}->{Cout}@{H} B{B} sample {i}", da[i:i + 1], ref, 1e-5)
    # weight gradient: from the V planes the forward left behind (production route) and by re-transforming x
    dw = torch.empty(9 * Cin * Cout, device="cuda")
    db = torch.empty(Cout, device="cuda")          # production entry: weight gradient + bias gradient from one pass over dy
    nbw = L.query("icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes", B, H, H, Cin, Cout, 36)
    L.call("icg_conv2d_wino4_wgrad_from_v_db", v_saved, dy, dw, db, B, H, H, Cin, Cout, 36, 0, 1.0, _bytes(nbw), nbw)
    _check(f"wino4 dbias (dy transform) {Cin}->{Cout}@{H} B{B} full batch", db, dy.double().sum((0, 2, 3)), 2e-5)
    dw2 = torch.empty_like(dw)
    nbw2 = L.query("icg_conv2d_wino4_wgrad_workspace_bytes", B, H, H, Cin, Cout)
    L.call("icg_conv2d_wino4_wgrad", x, dy, dw2, None, None, 0, B, H, H, Cin, Cout, PRE_RELU, _bytes(nbw2), nbw2)
    ref = torch.nn.grad.conv2d_weight(F.relu(x.cpu().contiguous()), (Cout, Cin, 3, 3), dy.cpu().contiguous(), padding=1)
    ref = ref.permute(2, 3, 1, 0).contiguous()                                  # HWIO
    _check(f"wino4 wgrad (saved V) {Cin}->{Cout}@{H} B{B} full batch", dw.view(3, 3, Cin, Cout), ref, 2e-5)
    _check(f"wino4 wgrad (re-transform) {Cin}->{Cout}@{H} B{B} full batch", dw2.view(3, 3, Cin, Cout), ref, 2e-5)


# ------------------------------------------------------------------------------------------------ upsample-fused, 25 planes
@pytest.mark.parametrize("B,Hs,Cin,Cout", [(64, 128, 192, 96), (64, 32, 768, 384)])
def test_upsample_fused_25plane_at_bench_shape(B, Hs, Cin, Cout):
    L = _L()
    import ic_gan_amd.ops as ops
    H = 2 * Hs
    assert ops.resample_winograd_applies(Cin, Cout, H, H, B) == 5
    w, wd, w_oihw = _weights(Cout, Cin, 4)
    x = gpu_randn_cl(B, Cin, Hs, Hs, 5)
    dy = gpu_randn_cl(B, Cout, H, H, 6)
    g = torch.Generator().manual_seed(7)
    sc, sh = 1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)
    flags = L.ICG_PRE_AFFINE | L.ICG_PRE_RELU
    U, Ud = torch.empty(25 * Cout * Cin, device="cuda"), torch.empty(25 * Cout * Cin, device="cuda")
    L.call("icg_wino4r_weight_transform", w.cuda(), U, Cout, Cin)
    L.call("icg_wino4r_weight_transform", wd.cuda(), Ud, Cin, Cout)
    bias = torch.linspace(-1, 1, Cout)
    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, H, Cin, Cout)
    ws = _bytes(nb)
    out = torch.empty_like(dy)
    L.call("icg_conv2d_up_wino_fprop", x, U, bias.cuda(), out, sc.cuda(), sh.cuda(), Cin, B, Hs, Hs, Cin, Cout, flags | L.ICG_WINO_KEEP_V, ws, nb)
    v_saved = ws[: 25 * B * (H // 4) * (H // 4) * Cin * 4].view(torch.float32)
    nbd = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, H, Cout, Cin)
    da = torch.empty_like(x)
    L.call("icg_conv2d_up_wino_dgrad", dy, Ud, da, B, Hs, Hs, Cin, Cout, _bytes(nbd), nbd)

    def act(i0, i1):
        a = x[i0:i1].cpu().contiguous() * sc[i0:i1, :, None, None] + sh[i0:i1, :, None, None]
        return F.interpolate(F.relu(a), scale_factor=2, mode="nearest")

    for i in _samples(B):
        _check(f"up25 fprop {Cin}->{Cout}@{Hs}->{H} B{B} sample {i}", out[i:i + 1], F.conv2d(act(i, i + 1), w_oihw, bias, padding=1), 1e-5)
        gi = torch.nn.grad.conv2d_input((1, Cin, H, H), w_oihw, dy[i:i + 1].cpu().contiguous(), padding=1)
        _check(f"up25 dgrad {Cin}->{Cout}@{Hs}->{H} B{B} sample {i}", da[i:i + 1], 4 * F.avg_pool2d(gi, 2), 1e-5)
    dw = torch.empty(9 * Cin * Cout, device="cuda")
    nbw = L.query("icg_conv2d_wino4_wgrad_from_v_workspace_bytes", B, H, H, Cin, Cout, 25)
    L.call("icg_conv2d_wino4_wgrad_from_v", v_saved, dy, dw, B, H, H, Cin, Cout, 25, 0, 1.0, _bytes(nbw), nbw)
    ref = torch.zeros(Cout, Cin, 3, 3)
    for i0 in range(0, B, 16):                         # chunks bound the host memory of the upsampled activation
        ref += torch.nn.grad.conv2d_weight(act(i0, i0 + 16), (Cout, Cin, 3, 3), dy[i0:i0 + 16].cpu().contiguous(), padding=1)
    _check(f"up25 wgrad (saved V) {Cin}->{Cout}@{Hs}->{H} B{B} full batch", dw.view(3, 3, Cin, Cout),
           ref.permute(2, 3, 1, 0).contiguous(), 2e-5)


# ------------------------------------------------------------------------------------------------ avgpool-fused, 25 planes
def test_avgpool_fused_25plane_at_bench_shape():
    L = _L()
    import ic_gan_amd.ops as ops
    B, H, Cin, Cout = 64, 256, 96, 192
    Hp = H // 2
    w, wd, w_oihw = _weights(Cout, Cin, 8)
    x = gpu_randn_cl(B, Cin, H, H, 9)
    dy = gpu_randn_cl(B, Cout, Hp, Hp, 10)
    res = gpu_randn_cl(B, Cout, Hp, Hp, 11)
    U, Ud = torch.empty(25 * Cout * Cin, device="cuda"), torch.empty(25 * Cout * Cin, device="cuda")
    L.call("icg_wino4r_weight_transform", w.cuda(), U, Cout, Cin)
    L.call("icg_wino4r_weight_transform", wd.cuda(), Ud, Cin, Cout)
    bias = torch.linspace(-1, 1, Cout)
    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, H, Cin, Cout)
    ws = _bytes(nb)
    out = torch.empty_like(dy)
    L.call("icg_conv2d_down_wino_fprop", x, U, bias.cuda(), res, out, B, Hp, Hp, Cin, Cout, PRE_RELU | L.ICG_WINO_KEEP_V, ws, nb)
    v_saved = ws[: 25 * B * (H // 4) * (H // 4) * Cin * 4].view(torch.float32)
    nbd = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, H, Cout, Cin)
    da = torch.empty_like(x)
    L.call("icg_conv2d_down_wino_dgrad", dy, Ud, da, B, Hp, Hp, Cin, Cout, _bytes(nbd), nbd)
    for i in _samples(B):
        xi = F.relu(x[i:i + 1].cpu().contiguous())
        ref = F.avg_pool2d(F.conv2d(xi, w_oihw, None, padding=1), 2) + bias[None, :, None, None] + res[i:i + 1].cpu()
        _check(f"down25 fprop {Cin}->{Cout}@{H}->{Hp} B{B} sample {i}", out[i:i + 1], ref, 1e-5)
        up = 0.25 * F.interpolate(dy[i:i + 1].cpu().contiguous(), scale_factor=2, mode="nearest")
        _check(f"down25 dgrad {Cin}->{Cout}@{H}->{Hp} B{B} sample {i}", da[i:i + 1],
               torch.nn.grad.conv2d_input((1, Cin, H, H), w_oihw, up, padding=1), 1e-5)
    dw = torch.empty(9 * Cin * Cout, device="cuda")
    db = torch.empty(Cout, device="cuda")
    nbw = L.query("icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes", B, H, H, Cin, Cout, 25)
    L.call("icg_conv2d_wino4_wgrad_from_v_db", v_saved, dy, dw, db, B, H, H, Cin, Cout, 25, 1, 0.25, _bytes(nbw), nbw)
    _check(f"down25 dbias (dy transform) {Cin}->{Cout}@{H}->{Hp} B{B} full batch", db, dy.double().sum((0, 2, 3)), 2e-5)
    ref = torch.zeros(Cout, Cin, 3, 3)
    for i0 in range(0, B, 16):
        up = 0.25 * F.interpolate(dy[i0:i0 + 16].cpu().contiguous(), scale_factor=2, mode="nearest")
        ref += torch.nn.grad.conv2d_weight(F.relu(x[i0:i0 + 16].cpu().contiguous()), (Cout, Cin, 3, 3), up, padding=1)
    _check(f"down25 wgrad (saved V) {Cin}->{Cout}@{H}->{Hp} B{B} full batch", dw.view(3, 3, Cin, Cout),
           ref.permute(2, 3, 1, 0).contiguous(), 2e-5)


# ------------------------------------------------------------------------------------------------ 64-bit offsets
def test_implicit_gemm_64bit_offsets():
    """2^31 input elements (8.6 GB): the loaders fall back from 32-bit element offsets (`small == false`)."""
    L = _L()
    B, H, Cin, Cout = 128, 256, 256, 32
    assert B * H * H * Cin >= 0x7fffffff
    w, wd, w_oihw = _weights(Cout, Cin, 12)
    x = gpu_randn_cl(B, Cin, H, H, 13)
    out = torch.empty(B, H, H, Cout, device="cuda").permute(0, 3, 1, 2)
    bias = torch.linspace(-1, 1, Cout)
    L.call("icg_conv2d_fprop", x, w.cuda(), bias.cuda(), None, out, None, None, 0, B, H, H, Cin, Cout, 3, PRE_RELU, 1.0)
    for i in _samples(B):
        xi = F.relu(x[i:i + 1].cpu().contiguous())
        _check(f"direct fprop 64-bit {Cin}->{Cout}@{H} B{B} sample {i}", out[i:i + 1], F.conv2d(xi, w_oihw, bias, padding=1), 1e-5)
    # weight gradient over the same 8.6 GB operand (K = 8.4M pixels, split-K)
    dy = gpu_randn_cl(B, Cout, H, H, 14)
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, H, Cin, Cout, 3)
    dw = torch.empty(9 * Cin * Cout, device="cuda")
    L.call("icg_conv2d_wgrad", x, dy, dw, None, None, 0, B, H, H, Cin, Cout, 3, PRE_RELU, _bytes(nb), nb)
    ref = torch.zeros(Cout, Cin, 3, 3)
    for i0 in range(0, B, 16):
        ref += torch.nn.grad.conv2d_weight(F.relu(x[i0:i0 + 16].cpu().contiguous()), (Cout, Cin, 3, 3),
                                           dy[i0:i0 + 16].cpu().contiguous(), padding=1)
    _check(f"direct wgrad 64-bit {Cin}->{Cout}@{H} B{B} full batch", dw.view(3, 3, Cin, Cout), ref.permute(2, 3, 1, 0).contiguous(), 2e-5)

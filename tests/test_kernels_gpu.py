"""GPU parity, kernel by kernel: every C-ABI entry point of libicgan_hip.so is run on the MI355X through the
product's binding (ic_gan_amd._lib.call) and compared with its plain-PyTorch fp32 CPU reference
(oracle/kernel_ref.py) on the same seeded inputs.  Tolerances are fp32-roundoff class and written per test
(north_star: generated samples within 1e-3 rel L2; kernels are held to ~1e-5)."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as R

pytestmark = pytest.mark.gpu

PRE_RELU, PRE_AFFINE, UP, RES_UP, RES_MASK = 1, 2, 4, 8, 16


def _L():
    import ic_gan_amd._lib as L
    return L


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def cl(b, c, h, w, seed=0, scale=1.0):
    return rnd(b, c, h, w, seed=seed, scale=scale).contiguous(memory_format=torch.channels_last)


def dev(t):
    return None if t is None else t.cuda()


def close(got, ref, rtol=2e-5, atol_rel=2e-5, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    tol = atol_rel * scale + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = int(torch.argmax(err - tol))
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} mismatches; worst at flat {idx}: got "
                             f"{got.flatten()[idx]:.7g} ref {ref.flatten()[idx]:.7g} (max|ref| {scale:.4g}); "
                             f"rel L2 {float((got - ref).norm() / (ref.norm() + 1e-30)):.3e}")


def run_pair(name, args, outs):
    """args: list of CPU tensors / scalars / None.  outs: indices of output tensors.  Runs the HIP kernel on
    device copies and the reference on the CPU tensors; returns [(gpu_out, ref_out), ...]."""
    L = _L()
    dargs = [a.cuda() if isinstance(a, torch.Tensor) else a for a in args]
    L.call(name, *dargs)
    torch.cuda.synchronize()
    getattr(R, name)(*args)
    return [(dargs[i], args[i]) for i in outs]


# ------------------------------------------------------------------------------------------------ conv / linear
CONV_CASES = [
    # B, H, W, Cin, Cout, R, flags, residual(0 none / 1 same / 2 half-res / 3 ReLU mask: ICG_RES_RELU_MASK), bias
    (2, 8, 8, 32, 32, 3, 0, 0, True),
    (2, 8, 8, 32, 48, 3, 0, 3, False),                   # data gradient of a ReLU-prologue layer: mask in the epilogue
    (1, 5, 7, 24, 20, 1, 0, 3, False),                   # ... 1x1, ragged N
    (2, 8, 8, 8, 16, 3, PRE_RELU, 1, True),              # K-tile straddles taps (Cin = 8)
    (1, 16, 16, 3, 96, 3, 0, 0, True),                   # RGB stem: thin-input direct kernels (narrow_conv.hip)
    (2, 9, 7, 3, 40, 3, 0, 0, False),                    # ... LP = 16 with 10 active lanes, odd sizes
    (1, 40, 5, 4, 256, 3, 0, 0, True),                   # ... LP = 64, two row segments, Cin = 4
    (2, 8, 8, 1, 16, 3, 0, 0, True),                     # ... single input channel
    (2, 9, 7, 3, 64, 3, 0, 0, True),                     # RGB stem on the MFMA form (9 Cin <= 32, Cout % 32 == 0): odd sizes, ragged last tile
    (1, 5, 33, 2, 32, 3, 0, 0, False),                   # ... two input channels, one column tile, odd width
    (3, 16, 16, 3, 128, 3, 0, 0, True),                  # ... four column tiles (BigGAN-deep stem width)
    (2, 12, 10, 1, 96, 3, 0, 0, True),                   # ... single input channel, three column tiles
    (2, 16, 16, 3, 96, 1, 0, 0, True),                   # 1x1 from-RGB shortcut (D block 0 conv_sc) on the same MFMA kernels: K = Cin
    (1, 9, 7, 2, 32, 1, 0, 0, False),                    # ... two input channels, odd sizes
    (1, 8, 8, 3, 96, 3, PRE_RELU, 0, True),              # prologue requested -> stays on the implicit-GEMM path
    (2, 16, 16, 96, 96, 3, PRE_AFFINE | PRE_RELU | UP, 2, True),   # GBlock conv1-like + half-res residual
    (2, 16, 16, 96, 96, 3, PRE_AFFINE | PRE_RELU, 2, True),
    (3, 6, 10, 16, 40, 3, PRE_AFFINE | PRE_RELU | UP, 0, False),  # non-square, ragged N
    (1, 4, 4, 256, 384, 3, PRE_RELU, 1, True),
    (2, 8, 8, 64, 48, 1, 0, 0, False),                   # attention theta
    (2, 8, 8, 192, 24, 1, 0, 0, False),
    (2, 16, 16, 96, 3, 3, PRE_AFFINE | PRE_RELU, 0, True),        # generator RGB tail (N = 3)
    (64, 1, 1, 657, 96, 1, 0, 0, False),                 # ccbn gain linear (K = 657): skinny linear kernels
    (100, 1, 1, 657, 1536, 1, 0, 0, True),               # two row groups, bias
    (7, 1, 1, 21, 20, 1, 0, 0, False),                   # N not a multiple of 8, K tail
    (300, 1, 1, 657, 96, 1, 0, 0, False),                # more than 256 rows: stays on the implicit-GEMM scalar path
    (64, 1, 1, 17, 256, 1, 0, 0, True),                  # z-chunk linear
    (4, 1, 1, 2048, 512, 1, 0, 0, True),                 # shared_feat
    (8, 1, 1, 128, 1, 1, 0, 0, True),                    # D output linear (N = 1)
    (64, 1, 1, 1536, 1, 1, 0, 0, True),                  # ... at a training batch: >= 16 rows -> row-streaming GEMM (smallm_nt_kernel), N = 1
    (128, 1, 1, 2048, 1536, 1, 0, 0, True),              # linear_feat at the D step's batch: 8 row groups, 16-byte loads
    (64, 1, 1, 96, 657, 1, 0, 0, False),                 # data gradient of a conditional-BN projection (N = 657)
    (17, 1, 1, 40, 22, 1, 0, 0, True),                   # ragged everything, two row groups
    (1, 32, 32, 32, 160, 3, 0, 0, True),                 # N = 160 -> TN = 4 with ragged last tile
    (5, 2, 2, 20, 20, 3, PRE_AFFINE, 0, True),           # tiny spatial, affine without relu
    (3, 7, 5, 16, 1, 3, 0, 0, False),                    # direct narrow-output kernels (narrow_conv.hip): LP = 4
    (1, 9, 9, 192, 4, 3, PRE_AFFINE | PRE_RELU, 0, True),  # LP = 64 with 48 active lanes, 4 outputs
    (2, 4, 6, 256, 2, 3, PRE_RELU, 0, True),             # LP = 64 full
    (2, 8, 8, 64, 3, 3, PRE_AFFINE, 0, False),           # LP = 16, affine without relu
    (2, 9, 7, 32, 1, 3, PRE_RELU, 0, True),              # to-RGB weight gradient on the MFMA form: one output channel, odd sizes
    (1, 40, 70, 64, 3, 3, PRE_AFFINE | PRE_RELU, 0, True),  # to-RGB forward on the MFMA form: two row segments, three column chunks (30 + 30 + 10)
    (2, 33, 31, 128, 2, 3, 0, 0, False),                 # ... four K groups, no prologue (data gradient of a from-RGB layer), ragged everything
    (1, 6, 10, 128, 2, 3, PRE_AFFINE | PRE_RELU, 0, False),  # ... two output channels, four column tiles
]


def _conv_inputs(case, seed):
    B, H, W, Cin, Cout, R, flags, res, bias = case
    up = 1 if flags & UP else 0
    x = cl(B, Cin, H >> up, W >> up, seed=seed)
    w = rnd(Cout, R, R, Cin, seed=seed + 1, scale=1.0 / np.sqrt(R * R * Cin))
    bvec = rnd(Cout, seed=seed + 2) if bias else None
    rflags = flags
    r = None
    if res == 1:
        r = cl(B, Cout, H, W, seed=seed + 3)
    elif res == 2:
        r = cl(B, Cout, H // 2, W // 2, seed=seed + 3)
        rflags |= RES_UP
    elif res == 3:
        r = cl(B, Cout, H, W, seed=seed + 3)
        rflags |= RES_MASK
    sc = sh = None
    ssb = 0
    if flags & PRE_AFFINE:
        sc = (1.0 + 0.3 * rnd(B, Cin, seed=seed + 4)).contiguous()
        sh = (0.3 * rnd(B, Cin, seed=seed + 5)).contiguous()
        ssb = Cin
    return x, w, bvec, r, sc, sh, ssb, rflags


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fprop(case):
    B, H, W, Cin, Cout, R, flags, res, bias = case
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs(case, 10)
    out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
    (pair,) = run_pair("icg_conv2d_fprop", [x, w, bvec, r, out, sc, sh, ssb, B, H, W, Cin, Cout, R, rflags, 1.0], [4])
    close(*pair, what=f"fprop {case}")


def test_conv2d_fprop_shared_affine_row():
    """scale/shift with a single row (plain bn: ss_bstride = 0)."""
    B, H, W, Cin, Cout, R = 3, 8, 8, 32, 32, 3
    x, w = cl(B, Cin, H, W, seed=1), rnd(Cout, R, R, Cin, seed=2, scale=0.06)
    sc, sh = 1 + 0.2 * rnd(Cin, seed=3), 0.2 * rnd(Cin, seed=4)
    out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
    (pair,) = run_pair("icg_conv2d_fprop", [x, w, None, None, out, sc, sh, 0, B, H, W, Cin, Cout, R,
                                            PRE_AFFINE | PRE_RELU, 1.0], [4])
    close(*pair, what="fprop shared affine")


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_dgrad_as_fprop(case):
    """data gradient = fprop on dy with the tap-flipped, transposed weight (no prologue)."""
    B, H, W, Cin, Cout, R, flags, res, bias = case
    w4 = rnd(Cout, Cin, R, R, seed=3, scale=1.0 / np.sqrt(R * R * Cin))
    wd = w4.flip(2, 3).permute(1, 2, 3, 0).contiguous()          # [Cin][R][R][Cout]
    dy = cl(B, Cout, H, W, seed=4)
    da = torch.empty(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
    L = _L()
    ddy, dwd, dda = dy.cuda(), wd.cuda(), da.cuda()
    L.call("icg_conv2d_fprop", ddy, dwd, None, None, dda, None, None, 0, B, H, W, Cout, Cin, R, 0, 1.0)
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), w4, dy.contiguous(), padding=R // 2)
    close(dda, ref.contiguous(memory_format=torch.channels_last), what=f"dgrad {case}")


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_wgrad(case):
    B, H, W, Cin, Cout, R, flags, res, bias = case
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs(case, 20)
    dy = cl(B, Cout, H, W, seed=31)
    dw = torch.empty(R * R * Cin * Cout)
    L = _L()
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, R)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    (pair,) = run_pair("icg_conv2d_wgrad", [x, dy, dw, sc, sh, ssb, B, H, W, Cin, Cout, R, flags, ws, nb], [2])
    close(*pair, rtol=5e-5, atol_rel=5e-5, what=f"wgrad {case}")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 32, 32, 96, 192), (2, 64, 64, 288, 384), (8, 32, 32, 64, 96), (1, 64, 64, 1536, 128),
                                            (4, 32, 32, 100, 96)])
def test_conv2d_wgrad_1x1_on_the_tn_plane_gemm(B, H, W, Cin, Cout):
    """prologue-free 1x1 weight gradients (block shortcuts, the attention projections) run on the second-generation TN plane GEMM with
    one plane (wgrad_1x1_tn_ok, csrc/gemm_conv.hip): against fp64, deterministic, and with the minimum workspace of the query."""
    L = _L()
    x, dy = cl(B, Cin, H, W, seed=1), cl(B, Cout, H, W, seed=2)
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, 1)
    dw = torch.empty(Cin * Cout)
    (pair,) = run_pair("icg_conv2d_wgrad", [x, dy, dw, None, None, 0, B, H, W, Cin, Cout, 1, 0, torch.empty(max(nb, 16), dtype=torch.uint8), nb], [2])
    close(*pair, rtol=5e-5, atol_rel=5e-5, what="1x1 wgrad")
    ref = (x.permute(0, 2, 3, 1).reshape(-1, Cin).double().t() @ dy.permute(0, 2, 3, 1).reshape(-1, Cout).double()).float().reshape(-1)
    close(pair[0], ref, rtol=2e-5, atol_rel=2e-5, what="1x1 wgrad vs fp64")
    a, b = torch.empty_like(dw).cuda(), torch.empty_like(dw).cuda()
    for o in (a, b):
        L.call("icg_conv2d_wgrad", x.cuda(), dy.cuda(), o, None, None, 0, B, H, W, Cin, Cout, 1, 0,
               torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda"), nb)
    assert torch.equal(a, b)


def test_conv2d_wgrad_split_k_large():
    """many pixels, few channels: exercises split-K + the deterministic slab reduction."""
    B, H, W, Cin, Cout, R = 4, 64, 64, 16, 24, 3
    x, dy = cl(B, Cin, H, W, seed=1), cl(B, Cout, H, W, seed=2)
    dw = torch.empty(R * R * Cin * Cout)
    L = _L()
    nb = L.query("icg_conv2d_wgrad_workspace_bytes", B, H, W, Cin, Cout, R)
    assert nb > 16, "expected a split-K plan"
    ws = torch.empty(nb, dtype=torch.uint8)
    (pair,) = run_pair("icg_conv2d_wgrad", [x, dy, dw, None, None, 0, B, H, W, Cin, Cout, R, PRE_RELU, ws, nb], [2])
    close(*pair, rtol=1e-4, atol_rel=1e-4, what="wgrad split-K")
    # determinism: two launches give bit-identical results
    a, b = torch.empty_like(dw).cuda(), torch.empty_like(dw).cuda()
    for o in (a, b):
        L.call("icg_conv2d_wgrad", x.cuda(), dy.cuda(), o, None, None, 0, B, H, W, Cin, Cout, R, PRE_RELU,
               ws.cuda(), nb)
    assert torch.equal(a, b)


UP_CASES = [  # B, Hs, Ws, Cin, Cout, flags
    (2, 8, 8, 32, 32, PRE_AFFINE | PRE_RELU), (2, 16, 16, 96, 96, PRE_AFFINE | PRE_RELU), (3, 3, 5, 16, 40, PRE_RELU),
    (1, 4, 4, 256, 128, PRE_AFFINE | PRE_RELU), (2, 8, 8, 8, 12, 0), (4, 4, 4, 64, 192, PRE_AFFINE | PRE_RELU),
]


def _phase_weights(w4, down=False):
    """[Cout,Cin,3,3] -> (wp [4][Cout][2][2][Cin], vd [Cin][4][4][Cout]) through the SN reference (sigma ~ 1 path)."""
    Cout, Cin = w4.shape[:2]
    n = Cout * Cin * 9
    v, uo, sg = torch.empty(Cin * 9), torch.empty(Cout), torch.empty(1)
    wo, wd, wp, vd = torch.empty(n), torch.empty(n), torch.empty(16 * Cout * Cin), torch.empty(16 * Cout * Cin)
    vdn, wq = torch.empty(16 * Cout * Cin), torch.empty(16 * Cout * Cin)
    R.icg_sn_forward(w4.clone(), rnd(1, Cout, seed=77), torch.ones(1), Cout, Cin, 3, 1e-6, 0, v, uo, sg, wo, wd, wp, vd,
                     vdn, wq, torch.empty(16, dtype=torch.uint8), 16)
    if down:
        return wo, vdn, wq
    return wo, wp, vd


@pytest.mark.parametrize("case", UP_CASES)
def test_conv2d_up_phase_triplet(case):
    """4-phase upsample-fused conv: fprop / dgrad / wgrad vs their references AND vs the direct formulation."""
    B, Hs, Ws, Cin, Cout, flags = case
    L = _L()
    w4 = rnd(Cout, Cin, 3, 3, seed=5, scale=1 / np.sqrt(9 * Cin))
    w_ohwi, wp, vd = _phase_weights(w4)
    x = cl(B, Cin, Hs, Ws, seed=6)
    bias = rnd(Cout, seed=7)
    sc = sh = None
    ssb = 0
    if flags & PRE_AFFINE:
        sc, sh, ssb = (1 + 0.3 * rnd(B, Cin, seed=8)).contiguous(), (0.3 * rnd(B, Cin, seed=9)).contiguous(), Cin
    out = torch.empty(B, Cout, 2 * Hs, 2 * Ws).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_up_fprop", [x, wp, bias, out, sc, sh, ssb, B, Hs, Ws, Cin, Cout, flags], [3])
    close(*p, what=f"up_fprop {case}")
    # equals the direct (upsampled-tensor) convolution
    direct = torch.empty_like(out)
    R.icg_conv2d_fprop(x, w_ohwi, bias, None, direct, sc, sh, ssb, B, 2 * Hs, 2 * Ws, Cin, Cout, 3, flags | UP, 1.0)
    close(p[0], direct, rtol=1e-4, atol_rel=2e-5, what=f"up_fprop vs direct {case}")
    # dgrad at source resolution == 2x2 sum of the direct data gradient
    dy = cl(B, Cout, 2 * Hs, 2 * Ws, seed=11)
    da = torch.empty(B, Cin, Hs, Ws).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_up_dgrad", [dy, vd, da, B, Hs, Ws, Cin, Cout], [2])
    close(*p, what=f"up_dgrad {case}")
    sigma_w = w_ohwi.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()
    full = torch.nn.grad.conv2d_input((B, Cin, 2 * Hs, 2 * Ws), sigma_w, dy.contiguous(), padding=1)
    ref = full.view(B, Cin, Hs, 2, Ws, 2).sum((3, 5)).contiguous(memory_format=torch.channels_last)
    close(p[0], ref, rtol=1e-4, atol_rel=2e-5, what=f"up_dgrad vs direct {case}")
    # wgrad in phase form
    nb = L.query("icg_conv2d_up_wgrad_workspace_bytes", B, Hs, Ws, Cin, Cout)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    dwp = torch.empty(16 * Cin * Cout)
    (p,) = run_pair("icg_conv2d_up_wgrad", [x, dy, dwp, sc, sh, ssb, B, Hs, Ws, Cin, Cout, flags, ws, nb], [2])
    close(*p, rtol=5e-5, atol_rel=5e-5, what=f"up_wgrad {case}")


def test_conv2d_up_wgrad_split_k():
    B, Hs, Ws, Cin, Cout = 4, 32, 32, 16, 24
    L = _L()
    x, dy = cl(B, Cin, Hs, Ws, seed=1), cl(B, Cout, 2 * Hs, 2 * Ws, seed=2)
    nb = L.query("icg_conv2d_up_wgrad_workspace_bytes", B, Hs, Ws, Cin, Cout)
    assert nb > 16
    ws = torch.empty(nb, dtype=torch.uint8)
    (p,) = run_pair("icg_conv2d_up_wgrad", [x, dy, torch.empty(16 * Cin * Cout), None, None, 0, B, Hs, Ws, Cin, Cout,
                                            PRE_RELU, ws, nb], [2])
    close(*p, rtol=1e-4, atol_rel=1e-4, what="up_wgrad split-K")


DOWN_CASES = [  # B, Hp, Wp, Cin, Cout, relu, residual
    (2, 8, 8, 32, 32, 1, 1), (2, 16, 16, 96, 96, 1, 1), (3, 3, 5, 16, 40, 0, 0), (1, 2, 2, 256, 128, 1, 1),
    (2, 4, 4, 8, 12, 1, 0), (4, 8, 8, 64, 192, 1, 1),
]


@pytest.mark.parametrize("case", DOWN_CASES)
def test_conv2d_down_fused_triplet(case):
    """conv3x3 -> avgpool2 as one 4x4/stride-2 conv: fprop / dgrad / wgrad vs references AND vs conv-then-pool."""
    B, Hp, Wp, Cin, Cout, relu, has_res = case
    L = _L()
    H, W = 2 * Hp, 2 * Wp
    flags = PRE_RELU if relu else 0
    w4 = rnd(Cout, Cin, 3, 3, seed=5, scale=1 / np.sqrt(9 * Cin))
    w_ohwi, vdn, wq = _phase_weights(w4, down=True)
    x = cl(B, Cin, H, W, seed=6)
    bias = rnd(Cout, seed=7)
    res = cl(B, Cout, Hp, Wp, seed=8) if has_res else None
    out = torch.empty(B, Cout, Hp, Wp).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_down_fprop", [x, vdn, bias, res, out, B, Hp, Wp, Cin, Cout, flags], [4])
    close(*p, what=f"down_fprop {case}")
    wn = w_ohwi.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()
    a = torch.relu(x) if relu else x
    direct = torch.nn.functional.avg_pool2d(torch.nn.functional.conv2d(a, wn, bias, padding=1), 2)
    if has_res:
        direct = direct + res
    close(p[0], direct.contiguous(memory_format=torch.channels_last), rtol=1e-4, atol_rel=2e-5,
          what=f"down_fprop vs conv+pool {case}")
    dy = cl(B, Cout, Hp, Wp, seed=11)
    da = torch.empty(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_down_dgrad", [dy, wq, da, B, Hp, Wp, Cin, Cout], [2])
    close(*p, what=f"down_dgrad {case}")
    up_dy = (0.25 * dy).repeat_interleave(2, 2).repeat_interleave(2, 3).contiguous()
    ref = torch.nn.grad.conv2d_input((B, Cin, H, W), wn, up_dy, padding=1)
    close(p[0], ref.contiguous(memory_format=torch.channels_last), rtol=1e-4, atol_rel=2e-5,
          what=f"down_dgrad vs direct {case}")
    dx = torch.empty(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
    (pm,) = run_pair("icg_conv2d_down_dgrad_relu", [dy, wq, x, dx, B, Hp, Wp, Cin, Cout], [3])
    close(*pm, what=f"down_dgrad_relu {case}")
    close(pm[0], torch.where(x > 0, p[0].cpu(), torch.zeros(())), rtol=0, atol_rel=0, what="down_dgrad_relu == mask(down_dgrad), bitwise")
    nb = L.query("icg_conv2d_down_wgrad_workspace_bytes", B, Hp, Wp, Cin, Cout)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    (p,) = run_pair("icg_conv2d_down_wgrad", [x, dy, torch.empty(16 * Cin * Cout), B, Hp, Wp, Cin, Cout, flags, ws, nb],
                    [2])
    close(*p, rtol=5e-5, atol_rel=5e-5, what=f"down_wgrad {case}")


RS_WINO_CASES = [
    # B, Hl, Wl (low resolution: source of the upsample / pooled output), Cin, Cout, flags
    (2, 4, 4, 32, 32, 0),
    (2, 8, 8, 64, 48, PRE_AFFINE | PRE_RELU),
    (1, 2, 6, 16, 40, PRE_RELU),              # one tile row, non-square
    (3, 16, 16, 128, 96, PRE_RELU),           # long K (tiles) -> batched split-K slabs in the weight gradient
]


def _rs_weights(Cin, Cout):
    """-> (w_ohwi [Cout][3][3][Cin], w_dgrad [Cin][3][3][Cout] with flipped taps, U25 of each)"""
    w = rnd(Cout, 3, 3, Cin, seed=5, scale=1 / np.sqrt(9 * Cin))
    wd = w.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    U, Ud = torch.empty(25 * Cout * Cin), torch.empty(25 * Cout * Cin)
    (pu,) = run_pair("icg_wino4r_weight_transform", [w, U, Cout, Cin], [1])
    close(*pu, what="wino4r weight transform")
    R.icg_wino4r_weight_transform(wd, Ud, Cin, Cout)
    return w, wd, U, Ud


@pytest.mark.parametrize("case", RS_WINO_CASES)
def test_conv2d_up_winograd_triplet(case):
    """upsample-fused 3x3 conv in the 25-plane F(4x4,3x3) domain: fprop / dgrad / wgrad vs the op graph evaluated directly."""
    B, Hs, Ws, Cin, Cout, flags = case
    L = _L()
    w, wd, U, Ud = _rs_weights(Cin, Cout)
    x = cl(B, Cin, Hs, Ws, seed=6)
    bias = rnd(Cout, seed=7)
    sc = sh = None
    ssb = 0
    if flags & PRE_AFFINE:
        sc, sh, ssb = (1 + 0.3 * rnd(B, Cin, seed=8)).contiguous(), (0.3 * rnd(B, Cin, seed=9)).contiguous(), Cin
    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, 2 * Hs, 2 * Ws, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8)
    out = torch.empty(B, Cout, 2 * Hs, 2 * Ws).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_up_wino_fprop", [x, U, bias, out, sc, sh, ssb, B, Hs, Ws, Cin, Cout, flags, ws, nb], [3])
    close(*p, rtol=2e-4, atol_rel=2e-4, what=f"up_wino_fprop {case}")
    dy = cl(B, Cout, 2 * Hs, 2 * Ws, seed=11)
    da = torch.empty(B, Cin, Hs, Ws).contiguous(memory_format=torch.channels_last)
    nbd = L.query("icg_conv2d_rs_wino_workspace_bytes", B, 2 * Hs, 2 * Ws, Cout, Cin)
    (p,) = run_pair("icg_conv2d_up_wino_dgrad", [dy, Ud, da, B, Hs, Ws, Cin, Cout, torch.empty(nbd, dtype=torch.uint8), nbd], [2])
    close(*p, rtol=2e-4, atol_rel=2e-4, what=f"up_wino_dgrad {case}")
    nbw = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, 2 * Hs, 2 * Ws, Cin, Cout)
    (p,) = run_pair("icg_conv2d_up_wino_wgrad", [x, dy, torch.empty(9 * Cin * Cout), sc, sh, ssb, B, Hs, Ws, Cin, Cout, flags,
                                                 torch.empty(nbw, dtype=torch.uint8), nbw], [2])
    close(*p, rtol=5e-4, atol_rel=5e-4, what=f"up_wino_wgrad {case}")


@pytest.mark.parametrize("case", RS_WINO_CASES)
@pytest.mark.parametrize("has_res", [False, True])
def test_conv2d_down_winograd_triplet(case, has_res):
    """conv3x3 -> avgpool2 in the 25-plane F(4x4,3x3) domain: fprop / dgrad / wgrad vs conv-then-pool."""
    B, Hp, Wp, Cin, Cout, flags = case
    flags &= PRE_RELU
    L = _L()
    w, wd, U, Ud = _rs_weights(Cin, Cout)
    H, W = 2 * Hp, 2 * Wp
    x = cl(B, Cin, H, W, seed=6)
    bias = rnd(Cout, seed=7)
    res = cl(B, Cout, Hp, Wp, seed=8) if has_res else None
    nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cin, Cout)
    out = torch.empty(B, Cout, Hp, Wp).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_down_wino_fprop", [x, U, bias, res, out, B, Hp, Wp, Cin, Cout, flags,
                                                   torch.empty(nb, dtype=torch.uint8), nb], [4])
    close(*p, rtol=2e-4, atol_rel=2e-4, what=f"down_wino_fprop {case}")
    dy = cl(B, Cout, Hp, Wp, seed=11)
    da = torch.empty(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
    nbd = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cout, Cin)
    (p,) = run_pair("icg_conv2d_down_wino_dgrad", [dy, Ud, da, B, Hp, Wp, Cin, Cout, torch.empty(nbd, dtype=torch.uint8), nbd], [2])
    close(*p, rtol=2e-4, atol_rel=2e-4, what=f"down_wino_dgrad {case}")
    dx = torch.empty(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
    (pm,) = run_pair("icg_conv2d_down_wino_dgrad_relu", [dy, Ud, x, dx, B, Hp, Wp, Cin, Cout, torch.empty(nbd, dtype=torch.uint8), nbd], [3])
    close(*pm, rtol=2e-4, atol_rel=2e-4, what=f"down_wino_dgrad_relu {case}")
    close(pm[0], torch.where(x > 0, p[0].cpu(), torch.zeros(())), rtol=0, atol_rel=0, what="down_wino_dgrad_relu == mask(down_wino_dgrad), bitwise")
    nbw = L.query("icg_conv2d_rs_wino_wgrad_workspace_bytes", B, H, W, Cin, Cout)
    (p,) = run_pair("icg_conv2d_down_wino_wgrad", [x, dy, torch.empty(9 * Cin * Cout), B, Hp, Wp, Cin, Cout, flags,
                                                   torch.empty(nbw, dtype=torch.uint8), nbw], [2])
    close(*p, rtol=5e-4, atol_rel=5e-4, what=f"down_wino_wgrad {case}")


@pytest.mark.parametrize("kind", ["plain", "up", "down"])
@pytest.mark.parametrize("case", [(2, 4, 4, 32, 32, 0), (2, 8, 8, 64, 48, PRE_RELU), (3, 16, 16, 128, 96, PRE_RELU)])
def test_conv2d_winograd4_wgrad_from_saved_v(case, kind):
    """the forward entries leave V = transform(act(x)) at the start of their workspace; the weight gradient computed from
    those planes equals the one that re-transforms x"""
    B, Hl, Wl, Cin, Cout, flags = case
    L = _L()
    w, wd, U25, _ = _rs_weights(Cin, Cout)
    H, W = 2 * Hl, 2 * Wl                                    # full resolution of the layer
    bias = rnd(Cout, seed=7)
    planes = 36 if kind == "plain" else 25
    if kind == "plain":
        U = torch.empty(36 * Cout * Cin)
        R.icg_wino4_weight_transform(w, U, Cout, Cin)
        x = cl(B, Cin, H, W, seed=6)
        nb = L.query("icg_conv2d_wino4_workspace_bytes", B, H, W, Cin, Cout)
        ws = torch.zeros(nb, dtype=torch.uint8)
        out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
        pairs = run_pair("icg_conv2d_wino4_fprop", [x, U, bias, None, out, None, None, 0, B, H, W, Cin, Cout, flags | 32, 1.0, ws, nb], [15])    # 32: ICG_WINO_KEEP_V
        dy = cl(B, Cout, H, W, seed=11)
        ref_args = ("icg_conv2d_wino4_wgrad", [x, dy, None, None, None, 0, B, H, W, Cin, Cout, flags, None, 0])
        dy_up, alpha = 0, 1.0
    elif kind == "up":
        x = cl(B, Cin, Hl, Wl, seed=6)
        nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cin, Cout)
        ws = torch.zeros(nb, dtype=torch.uint8)
        out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
        pairs = run_pair("icg_conv2d_up_wino_fprop", [x, U25, bias, out, None, None, 0, B, Hl, Wl, Cin, Cout, flags | 32, ws, nb], [13])
        dy = cl(B, Cout, H, W, seed=11)
        ref_args = ("icg_conv2d_up_wino_wgrad", [x, dy, None, None, None, 0, B, Hl, Wl, Cin, Cout, flags, None, 0])
        dy_up, alpha = 0, 1.0
    else:
        x = cl(B, Cin, H, W, seed=6)
        nb = L.query("icg_conv2d_rs_wino_workspace_bytes", B, H, W, Cin, Cout)
        ws = torch.zeros(nb, dtype=torch.uint8)
        out = torch.empty(B, Cout, Hl, Wl).contiguous(memory_format=torch.channels_last)
        pairs = run_pair("icg_conv2d_down_wino_fprop", [x, U25, bias, None, out, B, Hl, Wl, Cin, Cout, flags | 32, ws, nb], [11])
        dy = cl(B, Cout, Hl, Wl, seed=11)
        ref_args = ("icg_conv2d_down_wino_wgrad", [x, dy, None, B, Hl, Wl, Cin, Cout, flags, None, 0])
        dy_up, alpha = 1, 0.25
    nv = planes * B * (H // 4) * (W // 4) * Cin
    (gws, cws), = pairs
    v_gpu, v_cpu = gws.view(torch.float32)[:nv], cws.view(torch.float32)[:nv]
    close(v_gpu, v_cpu, rtol=2e-5, atol_rel=2e-5, what=f"V planes left by the {kind} forward {case}")
    nbw = L.query("icg_conv2d_wino4_wgrad_from_v_workspace_bytes", B, H, W, Cin, Cout, planes)
    dw = torch.empty(9 * Cin * Cout)
    (p,) = run_pair("icg_conv2d_wino4_wgrad_from_v", [v_cpu.clone(), dy, dw, B, H, W, Cin, Cout, planes, dy_up, alpha,
                                                      torch.empty(nbw, dtype=torch.uint8), nbw], [2])
    close(*p, rtol=5e-4, atol_rel=5e-4, what=f"wgrad from saved V {kind} {case}")
    name, a = ref_args                      # ... and equals the reference of the entry that re-transforms x
    dw_ref = torch.empty(9 * Cin * Cout)
    a[2] = dw_ref
    getattr(R, name)(*a)
    close(p[0], dw_ref, rtol=5e-4, atol_rel=5e-4, what=f"wgrad from saved V vs re-transform {kind} {case}")
    # ... and the variant that also emits the bias gradient (column sums of dy) from the dy-transform pass
    nbd = L.query("icg_conv2d_wino4_wgrad_from_v_db_workspace_bytes", B, H, W, Cin, Cout, planes)
    (p2, pb) = run_pair("icg_conv2d_wino4_wgrad_from_v_db", [v_cpu.clone(), dy, torch.empty(9 * Cin * Cout), torch.empty(Cout), B, H, W,
                                                              Cin, Cout, planes, dy_up, alpha, torch.empty(nbd, dtype=torch.uint8), nbd],
                        [2, 3])
    assert torch.equal(p2[0].cpu(), p[0].cpu()), "the dbias variant must not change dw"
    close(*pb, rtol=2e-5, atol_rel=2e-5, what=f"dbias from the dy transform {kind} {case}")


GEMM_CASES = [
    # M, N, K, transA, transB, batch
    (256, 64, 4, 0, 1, 3), (256, 16, 64, 0, 0, 3), (64, 16, 256, 1, 0, 3), (64, 4, 256, 1, 0, 2),
    (1024, 256, 48, 0, 1, 2), (1024, 192, 256, 0, 0, 2), (256, 192, 1024, 1, 0, 2), (256, 24, 1024, 1, 0, 2),
    (130, 70, 24, 0, 1, 1), (100, 36, 52, 0, 0, 1), (36, 20, 100, 1, 0, 2),
    # a handful of rows, one GEMM (StyleGAN2's dense layers at batch 16): the row-streaming kernel (gemm_conv.hip: smallm_nt_kernel)
    (16, 512, 512, 0, 1, 1), (16, 512, 8192, 0, 1, 1), (5, 36, 132, 0, 1, 1), (32, 70, 1024, 0, 1, 1), (17, 16, 128, 0, 1, 1),
]


@pytest.mark.parametrize("case", GEMM_CASES)
def test_gemm_batched(case):
    M, N, K, ta, tb, batch = case
    A = rnd(batch, M * K, seed=1, scale=1 / np.sqrt(K))
    Bm = rnd(batch, N * K, seed=2)
    C = torch.empty(batch, M * N)
    (pair,) = run_pair("icg_gemm_batched", [A, Bm, C, M, N, K, ta, tb, M * K, N * K, M * N, batch, 0.5], [2])
    close(*pair, what=f"gemm {case}")


# the batched GEMM over Winograd planes as the composites launch it (icg_plane_gemm): second-generation kernel (csrc/pgemm.hip:
# LDS-DMA staging, 16-byte fragments) where N % 128 == 0 or N % 96 == 0 and K % 32 == 0, first-generation kernels otherwise
PLANE_GEMM_CASES = [
    # M (tiles), N (Cout), K (Cin), planes
    (256, 128, 32, 3),            # shortest K the DMA kernel takes: 2 K-tiles = prologue + over-fetch only
    (128, 128, 64, 2),            # exactly one m-tile
    (1, 128, 96, 2),              # a single row: 127 clamped rows per tile
    (130, 256, 96, 5),            # ragged M (2 rows in the second m-tile), two n-tiles, odd plane count
    (1000, 96, 96, 4),            # 96-column tile (12 B rows per wave, lanes 48..63 masked in the B DMA), ragged M
    (384, 192, 192, 36),          # 96-column tile, two n-tiles, F(4x4,3x3) plane count
    (257, 384, 128, 25),          # 128-column tile with three n-tiles, 25-plane count, K = 128 (8 K-tiles: not a multiple of 6)
    (512, 768, 1536, 2),          # longest K of the product (96 K-tiles)
    (4096, 128, 384, 16),         # > 16 workgroups per plane: XCD-aware order, several tiles per XCD
    (300, 160, 96, 3),            # N takes neither tile -> first-generation persistent body
    (300, 128, 48, 3),            # K % 32 != 0 -> first generation
]


@pytest.mark.parametrize("case", PLANE_GEMM_CASES)
def test_plane_gemm(case):
    M, N, K, planes = case
    A = rnd(planes, M * K, seed=11, scale=1 / np.sqrt(K))
    Bm = rnd(planes, N * K, seed=12)
    C = torch.full((planes, M * N), float("nan"))
    (pair,) = run_pair("icg_plane_gemm", [A, Bm, C, M, N, K, planes, 0.75], [2])
    close(*pair, what=f"plane gemm {case}")
    # against fp64: two-level accumulation keeps the chain error below the plain fp32 reference's own
    ref64 = 0.75 * torch.bmm(A.view(planes, M, K).double(), Bm.view(planes, N, K).double().transpose(1, 2))
    got = pair[0].cpu().double().view(planes, M, N)
    rel = float((got - ref64).norm() / ref64.norm())
    assert rel < 3e-6, rel


PLANE_GEMM_TN_CASES = [
    # M (Cin), N (Cout), K (tiles), planes
    (96, 96, 64, 2),              # one slice, 96-column tile (lanes 48..63 masked in the B DMA), M < 128: clamped columns
    (128, 128, 32, 3),            # shortest K: prologue + over-fetch only
    (192, 384, 4096, 4),          # several slices, ragged last m-tile (192 = 128 + 64), three n-tiles
    (384, 192, 2048, 25),         # 25 planes, 96-column tiles
    (768, 768, 1024, 2),          # many output tiles per plane
    (96, 192, 131072, 2),         # the product's longest K (D block 1 conv1 at B = 128): many slices
    (192, 192, 2048, 3),          # 96-row tiles (M a multiple of 96, not of 128): two m-tiles, two n-tiles, 6 waves per workgroup
    (192, 96, 4096, 25),          # ... 25 planes
    (288, 96, 1024, 2),           # ... three m-tiles
    (100, 96, 4096, 2),           # M % 4 == 0 but not a multiple of 16
    (96, 160, 4096, 2),           # N takes neither tile -> first generation
    (96, 96, 4112, 2),            # K % 32 != 0 -> first generation
]


@pytest.mark.parametrize("case", PLANE_GEMM_TN_CASES)
def test_plane_gemm_tn(case):
    M, N, K, planes = case
    L = _L()
    A = rnd(planes, K * M, seed=13)
    Bm = rnd(planes, K * N, seed=14, scale=1 / np.sqrt(K))
    nb = L.query("icg_plane_gemm_tn_workspace_bytes", M, N, K, planes)
    C = torch.full((planes, M * N), float("nan"))
    Ad, Bd, Cd = A.cuda(), Bm.cuda(), C.cuda()
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    outs = []
    for _ in range(2):
        Cd.fill_(float("nan"))
        L.call("icg_plane_gemm_tn", Ad, Bd, Cd, M, N, K, planes, ws, nb)
        torch.cuda.synchronize()
        outs.append(Cd.clone())
    assert torch.equal(outs[0], outs[1])
    ref64 = torch.bmm(A.view(planes, K, M).double().transpose(1, 2), Bm.view(planes, K, N).double())
    got = outs[0].cpu().double().view(planes, M, N)
    assert torch.isfinite(got).all()
    rel = float((got - ref64).norm() / ref64.norm())
    assert rel < 3e-6, rel
    close(outs[0].cpu(), ref64.float().reshape(planes, M * N), rtol=1e-4, atol_rel=2e-5, what=f"plane gemm tn {case}")


def test_plane_gemm_is_deterministic_and_leaves_neighbours_alone():
    """same launch twice -> bit-identical (no race between the DMA ring and the fragment reads shows up as run-to-run
    differences); rows / planes beyond the problem are not written (ragged M, guard regions around C)."""
    L = _L()
    M, N, K, planes = 333, 256, 192, 7
    A = rnd(planes, M * K, seed=21).cuda()
    Bm = rnd(planes, N * K, seed=22).cuda()
    guard = 4096
    buf = torch.full((guard + planes * M * N + guard,), 7.5, device="cuda")
    C = buf[guard: guard + planes * M * N]
    outs = []
    for _ in range(3):
        C.fill_(float("nan"))
        L.call("icg_plane_gemm", A, Bm, C, M, N, K, planes, 1.0)
        torch.cuda.synchronize()
        outs.append(C.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all()
    assert bool((buf[:guard] == 7.5).all()) and bool((buf[guard + planes * M * N:] == 7.5).all())


# ------------------------------------------------------------------------------------------------ batch norm
BN_SHAPES = [(128, 8), (4096, 96), (70001, 192), (64, 1536), (100, 384), (3, 4), (20000, 32)]


@pytest.mark.parametrize("rows,C", BN_SHAPES)
def test_bn_forward_chain(rows, C):
    """partial stats -> reduce -> finalize vs the fp64 reference; running stats, scale/shift."""
    L = _L()
    B = 4
    x = rnd(rows, C, seed=3) * (1 + torch.arange(C).float() / C) + 3.0 * rnd(C, seed=4)   # non-trivial means
    rm, rv = 0.1 * rnd(C, seed=5), 1 + 0.2 * torch.rand(C, generator=torch.Generator().manual_seed(6))
    gain, beta = 0.3 * rnd(B, C, seed=7), 0.3 * rnd(B, C, seed=8)
    outs = {}
    for tag, to in (("gpu", lambda t: t.cuda()), ("ref", lambda t: t.clone())):
        xx, rmm, rvv = to(x), to(rm), to(rv)
        nb = L.query("icg_bn_workspace_bytes", rows, C)
        ws = to(torch.empty(max(nb, 2 * C * 4, 16), dtype=torch.uint8))
        sums = to(torch.empty(2 * C, dtype=torch.float64))
        mean, invstd = to(torch.empty(C)), to(torch.empty(C))
        scale, shift = to(torch.empty(B, C)), to(torch.empty(B, C))
        fn = (lambda n, *a: L.call(n, *a)) if tag == "gpu" else (lambda n, *a: getattr(R, n)(*a))
        fn("icg_bn_partial_stats", xx, rmm, rows, C, ws, ws.numel())
        fn("icg_bn_reduce_partials", ws, rows, C, sums)
        fn("icg_bn_finalize", sums, rmm, float(rows), rmm, rvv, 0.1, 1e-5, 1, to(gain), to(beta), B, 1.0, C, mean,
           invstd, scale, shift)
        outs[tag] = dict(sums=sums, mean=mean, invstd=invstd, scale=scale, shift=shift, rm=rmm, rv=rvv)
        if tag == "gpu":
            # the one-launch form of reduce + finalize: bit for bit the two-kernel path (same partials)
            rm2, rv2 = to(rm), to(rv)
            m2, i2, sc2, sh2 = (torch.empty_like(t) for t in (mean, invstd, scale, shift))
            fn("icg_bn_reduce_finalize", ws, rows, C, rm2, rm2, rv2, 0.1, 1e-5, to(gain), to(beta), B, 1.0, m2, i2, sc2, sh2)
            for a, b, what in ((m2, mean, "mean"), (i2, invstd, "invstd"), (sc2, scale, "scale"), (sh2, shift, "shift"), (rm2, rmm, "rm"),
                               (rv2, rvv, "rv")):
                assert torch.equal(a, b), what
    for k in outs["gpu"]:
        close(outs["gpu"][k], outs["ref"][k], rtol=3e-5, atol_rel=3e-5, what=f"bn fwd {k} {rows}x{C}")
    # against torch's own batch_norm statistics
    ref_mean, ref_var = x.double().mean(0), x.double().var(0, unbiased=False)
    close(outs["gpu"]["mean"], ref_mean.float(), rtol=1e-5, atol_rel=1e-5, what="bn mean vs torch")
    close(outs["gpu"]["invstd"], (1 / torch.sqrt(ref_var + 1e-5)).float(), rtol=1e-4, atol_rel=1e-5, what="bn invstd")


def test_bn_finalize_eval_mode():
    L = _L()
    C, B = 96, 3
    rm, rv = rnd(C, seed=1), 1 + torch.rand(C, generator=torch.Generator().manual_seed(2))
    gain = rnd(1, C, seed=3)
    res = {}
    for tag, to in (("gpu", lambda t: t.cuda()), ("ref", lambda t: t.clone())):
        mean, invstd, scale, shift = (to(torch.empty(C)) for _ in range(4))
        rmm, rvv = to(rm), to(rv)
        fn = (lambda n, *a: L.call(n, *a)) if tag == "gpu" else (lambda n, *a: getattr(R, n)(*a))
        fn("icg_bn_finalize", None, None, 0.0, rmm, rvv, 0.1, 1e-5, 0, to(gain), None, 1, 0.0, C, mean, invstd, scale,
           shift)
        res[tag] = (mean, invstd, scale, shift, rmm, rvv)
    for a, b in zip(res["gpu"], res["ref"]):
        close(a, b, what="bn finalize eval")


BNB_CASES = [(2, 8, 8, 16, PRE_AFFINE | PRE_RELU, 2), (2, 8, 8, 96, PRE_AFFINE | PRE_RELU | UP, 2),
             (3, 5, 7, 32, PRE_AFFINE | PRE_RELU, 3), (4, 4, 4, 1536, PRE_AFFINE | PRE_RELU | UP, 4),
             (2, 16, 16, 192, PRE_AFFINE, 1), (2, 32, 32, 24, PRE_AFFINE | PRE_RELU | UP, 1)]


@pytest.mark.parametrize("B,Hs,Ws,C,flags,gb_rows", BNB_CASES)
def test_bn_backward_chain(B, Hs, Ws, C, flags, gb_rows):
    L = _L()
    up = 2 if flags & UP else 1
    x = cl(B, C, Hs, Ws, seed=1)
    da = cl(B, C, Hs * up, Ws * up, seed=2)
    rows = gb_rows if gb_rows > 1 else 1
    gb = B if gb_rows > 1 else 1
    gain = 0.3 * rnd(gb, C, seed=3)
    mean, invstd = 0.2 * rnd(C, seed=4), 1 + 0.3 * torch.rand(C, generator=torch.Generator().manual_seed(5))
    scale = (invstd.view(1, C) * (1.0 + gain)).contiguous()
    shift = (0.2 * rnd(gb, C, seed=6) - mean.view(1, C) * scale).contiguous()
    ssb = C if gb > 1 else 0
    res = {}
    for tag, to in (("gpu", lambda t: t.cuda()), ("ref", lambda t: t.clone())):
        fn = (lambda n, *a: L.call(n, *a)) if tag == "gpu" else (lambda n, *a: getattr(R, n)(*a))
        nb = L.query("icg_bn_bwd_workspace_bytes", B, Hs, Ws, C)
        ws = to(torch.empty(max(nb, 16), dtype=torch.uint8))
        sd, sx = to(torch.empty(B, C)), to(torch.empty(B, C))
        xx, dd, sc, sh, mu, istd, gg = to(x), to(da), to(scale), to(shift), to(mean), to(invstd), to(gain)
        fn("icg_bn_bwd_reduce", xx, dd, sc, sh, ssb, mu, B, Hs, Ws, C, flags, ws, ws.numel(), sd, sx)
        chan = to(torch.empty(2 * C, dtype=torch.float64))
        fn("icg_bn_bwd_channel_sums", sd, sx, gg, gb, 1.0, istd, B, C, chan)
        dgain, dbeta = to(torch.empty(gb, C)), to(torch.empty(gb, C))
        ca, cb = to(torch.empty(C)), to(torch.empty(C))
        fn("icg_bn_bwd_coefs", sd, sx, chan, istd, float(B * Hs * Ws), 1, gb, B, C, dgain, dbeta, ca, cb)
        dx = to(torch.empty(B, C, Hs, Ws).contiguous(memory_format=torch.channels_last))
        fn("icg_bn_bwd_apply", xx, dd, sc, sh, ssb, mu, ca, cb, B, Hs, Ws, C, flags, dx)
        res[tag] = dict(sd=sd, sx=sx, chan=chan, dgain=dgain, dbeta=dbeta, ca=ca, cb=cb, dx=dx)
    for k in res["gpu"]:
        close(res["gpu"][k], res["ref"][k], rtol=5e-5, atol_rel=5e-5, what=f"bn bwd {k}")


def test_bn_backward_matches_autograd():
    """the four-stage backward equals autograd through  relu(batch_norm(x)*(1+g)+b)  followed by nearest x2."""
    L = _L()
    B, C, Hs, Ws = 3, 32, 6, 6
    x = cl(B, C, Hs, Ws, seed=1)
    g, be = 0.3 * rnd(B, C, seed=2), 0.3 * rnd(B, C, seed=3)
    da = cl(B, C, 2 * Hs, 2 * Ws, seed=4)
    xr, gr, br = x.clone().double().requires_grad_(True), g.clone().double().requires_grad_(True), \
        be.clone().double().requires_grad_(True)
    y = torch.nn.functional.batch_norm(xr, None, None, None, None, True, 0.1, 1e-5)
    a = torch.relu(y * (1 + gr).view(B, C, 1, 1) + br.view(B, C, 1, 1))
    a = torch.nn.functional.interpolate(a, scale_factor=2)
    a.backward(da.double())
    import ic_gan_amd.ops as ops
    bn = ops.BNOpt(torch.zeros(C).cuda(), torch.ones(C).cuda(), 1e-5, 0.1, True, 1.0, None)
    xc = x.cuda().requires_grad_(True)
    gc, bc = g.cuda().requires_grad_(True), be.cuda().requires_grad_(True)
    out = ops.norm_act(xc, bn, gc, bc, relu=True)
    ref_fwd = torch.relu(torch.nn.functional.batch_norm(x.double(), None, None, None, None, True, 0.1, 1e-5)
                         * (1 + g.double()).view(B, C, 1, 1) + be.double().view(B, C, 1, 1))
    close(out, ref_fwd.float(), rtol=1e-4, atol_rel=1e-5, what="norm_act fwd")
    # backward with the upsample adjoint folded (sum-pool da first, then stand-alone backward)
    dsum = da.view(B, C, Hs, 2, Ws, 2).sum((3, 5)).contiguous(memory_format=torch.channels_last)
    out.backward(dsum.cuda())
    close(xc.grad, xr.grad.float(), rtol=2e-4, atol_rel=2e-5, what="bn dx vs autograd")
    close(gc.grad, gr.grad.float(), rtol=2e-4, atol_rel=2e-5, what="bn dgain vs autograd")
    close(bc.grad, br.grad.float(), rtol=2e-4, atol_rel=2e-5, what="bn dbias vs autograd")


# ------------------------------------------------------------------------------------------------ spectral norm
SN_CASES = [(32, 32, 3), (3, 96, 3), (96, 3, 3), (10, 16, 1), (1000, 64, 1), (384, 384, 3), (24, 192, 1), (1, 128, 1),
            (512, 30, 1)]


@pytest.mark.parametrize("rows,Cin,taps", SN_CASES)
def test_sn_forward_backward(rows, Cin, taps):
    L = _L()
    w = rnd(rows, Cin, taps, taps, seed=1, scale=1 / np.sqrt(Cin * taps * taps))
    u = rnd(1, rows, seed=2)
    dw_hwio = rnd(taps * taps * Cin * rows, seed=3)
    n = rows * Cin * taps * taps
    res = {}
    for tag, to in (("gpu", lambda t: t.cuda()), ("ref", lambda t: t.clone())):
        fn = (lambda nm, *a: L.call(nm, *a)) if tag == "gpu" else (lambda nm, *a: getattr(R, nm)(*a))
        ww, uu, sv = to(w), to(u), to(torch.ones(1))
        v, uo, sg = to(torch.empty(Cin * taps * taps)), to(torch.empty(rows)), to(torch.empty(1))
        wo, wd = to(torch.empty(n)), to(torch.empty(n))
        wup = to(torch.empty(16 * rows * Cin)) if taps == 3 else None
        wupd = to(torch.empty(16 * rows * Cin)) if taps == 3 else None
        wdn = to(torch.empty(16 * rows * Cin)) if taps == 3 else None
        wdnd = to(torch.empty(16 * rows * Cin)) if taps == 3 else None
        nb = max(L.query("icg_sn_scratch_bytes", rows, Cin, taps), L.query("icg_sn_backward_scratch_bytes", rows, Cin, taps))
        sc = to(torch.empty(max(nb, 4096), dtype=torch.uint8))
        fn("icg_sn_forward", ww, uu, sv, rows, Cin, taps, 1e-6, 1, v, uo, sg, wo, wd, wup, wupd, wdn, wdnd, sc,
           sc.numel())
        dw = to(torch.empty(rows, Cin, taps, taps))
        fn("icg_sn_backward", to(dw_hwio), None, None, None, wo, uo, v, sg, rows, Cin, taps, dw, 0, sc, sc.numel())
        res[tag] = dict(u=uu, sv=sv, v=v, uo=uo, sigma=sg, w_ohwi=wo, w_dgrad=wd, dw=dw)
        if taps == 3:
            dw2 = to(torch.empty(rows, Cin, taps, taps))
            fn("icg_sn_backward", None, None, to(rnd(16 * rows * Cin, seed=9)), None, wo, uo, v, sg, rows, Cin, taps,
               dw2, 0, sc, sc.numel())
            dw3 = to(torch.empty(rows, Cin, taps, taps))
            fn("icg_sn_backward", None, None, None, to(rnd(16 * rows * Cin, seed=10)), wo, uo, v, sg, rows, Cin, taps,
               dw3, 0, sc, sc.numel())
            res[tag].update(w_up=wup, w_up_dgrad=wupd, dw_from_up=dw2, w_down=wdn, w_down_dgrad=wdnd, dw_from_down=dw3)
    for k in res["gpu"]:
        close(res["gpu"][k], res["ref"][k], rtol=5e-5, atol_rel=5e-5, what=f"sn {k} {rows}x{Cin}x{taps}")


# ------------------------------------------------------------------------------------------------ pointwise / pooling
@pytest.mark.parametrize("B,C,H,W", [(2, 3, 8, 8), (2, 96, 16, 16), (1, 5, 6, 10), (3, 48, 4, 4)])
def test_layout_pool_pointwise(B, C, H, W):
    x = cl(B, C, H, W, seed=1)
    n = x.numel()
    # layout
    xn = rnd(B, C, H, W, seed=2)
    y = torch.empty(B * C * H * W)
    (p,) = run_pair("icg_nchw_to_nhwc", [xn, y, B, C, H, W], [1]); close(*p, what="nchw->nhwc")
    (p,) = run_pair("icg_nhwc_to_nchw", [y.clone(), torch.empty(n), B, C, H, W], [1]); close(*p, what="nhwc->nchw")
    # pooling
    add = cl(B, C, H // 2, W // 2, seed=3)
    yp = torch.empty_like(add)
    (p,) = run_pair("icg_avgpool2_fwd", [x, add, yp, B, H, W, C], [2]); close(*p, what="avgpool+add")
    (p,) = run_pair("icg_avgpool2_fwd", [x, None, yp.clone(), B, H, W, C], [2]); close(*p, what="avgpool")
    (p,) = run_pair("icg_sumpool2_fwd", [x, yp.clone(), B, H, W, C], [1]); close(*p, what="sumpool")
    (p,) = run_pair("icg_avgpool2_bwd", [add, torch.empty_like(x), B, H, W, C], [1]); close(*p, what="avgpool bwd")
    (p,) = run_pair("icg_avgpool2_bwd_add", [add, cl(B, C, H, W, seed=13), torch.empty_like(x), B, H, W, C], [2]); close(*p, what="avgpool bwd + carry")
    (p,) = run_pair("icg_maxpool2_fwd", [x, yp.clone(), B, H, W, C], [1]); close(*p, what="maxpool")
    (p,) = run_pair("icg_maxpool2_bwd", [x, add, torch.empty_like(x), B, H, W, C], [2]); close(*p, what="maxpool bwd")
    # pointwise
    (p,) = run_pair("icg_tanh_fwd", [x, torch.empty_like(x), n], [1]); close(*p, what="tanh")
    (p,) = run_pair("icg_tanh_bwd", [torch.tanh(x), cl(B, C, H, W, seed=5), torch.empty_like(x), n], [2])
    close(*p, what="tanh bwd")
    (p,) = run_pair("icg_relu_fwd", [x, torch.empty_like(x), n], [1]); close(*p, what="relu")
    (p,) = run_pair("icg_relu_bwd", [x, cl(B, C, H, W, seed=6), torch.empty_like(x), n], [2]); close(*p, what="relu bwd")
    (p,) = run_pair("icg_add", [x, cl(B, C, H, W, seed=7), torch.empty_like(x), n], [2]); close(*p, what="add")
    # relu + sum pool
    (p,) = run_pair("icg_relu_sumpool_fwd", [x, torch.empty(B, C), B, H * W, C], [1]); close(*p, what="relu sumpool")
    (p,) = run_pair("icg_relu_sumpool_bwd", [x, rnd(B, C, seed=8), torch.empty_like(x), B, H * W, C], [2])
    close(*p, what="relu sumpool bwd")
    # gamma * o + x
    gamma = torch.tensor([0.37])
    (p,) = run_pair("icg_scale_add_fwd", [gamma, x, cl(B, C, H, W, seed=9), torch.empty_like(x), n], [3])
    close(*p, what="scale_add")
    sc = torch.empty(4096, dtype=torch.uint8)
    ps = run_pair("icg_scale_add_bwd", [gamma, x, cl(B, C, H, W, seed=10), torch.empty_like(x), torch.empty(1), n, sc,
                                        4096], [3, 4])
    close(*ps[0], what="scale_add d_o"); close(*ps[1], rtol=1e-4, atol_rel=1e-5, what="scale_add dgamma")


@pytest.mark.parametrize("rows,cols", [(64, 16), (1000, 64), (4096, 256), (333, 1024), (7, 1000), (50, 512), (9, 2048), (5, 1280)])
def test_softmax(rows, cols):
    x = rnd(rows, cols, seed=1, scale=3.0)
    (p,) = run_pair("icg_softmax_fwd", [x, torch.empty_like(x), rows, cols], [1]); close(*p, what="softmax")
    y = torch.softmax(x, -1)
    (p,) = run_pair("icg_softmax_bwd", [y, rnd(rows, cols, seed=2), torch.empty_like(x), rows, cols], [2])
    close(*p, what="softmax bwd")


@pytest.mark.parametrize("rows,C", [(4096, 96), (128, 1536), (70000, 3), (64, 1), (5000, 48), (33, 657), (16, 384), (300000, 192), (1000, 4), (123, 256), (77, 12)])
def test_colsum(rows, C):
    L = _L()
    x = rnd(rows, C, seed=1) + 0.5
    nb = L.query("icg_colsum_workspace_bytes", rows, C)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8)
    (p,) = run_pair("icg_colsum", [x, rows, C, torch.empty(C), ws, ws.numel()], [3])
    close(*p, rtol=1e-4, atol_rel=1e-5, what=f"colsum {rows}x{C}")


# ------------------------------------------------------------------------------------------------ optimiser / EMA
def test_adam_and_ema_multi():
    import ic_gan_amd.ops as ops
    sizes = [1, 3, 17, 4096, 4097, 100000, 5, 64 * 64 * 9] + [7 + i for i in range(60)]   # > 48 tensors: 2 launches
    ps = [rnd(n, seed=i) for i, n in enumerate(sizes)]
    gs = [rnd(n, seed=100 + i, scale=0.1) for i, n in enumerate(sizes)]
    ms = [0.01 * rnd(n, seed=200 + i) for i, n in enumerate(sizes)]
    vs = [(0.01 * rnd(n, seed=300 + i)) ** 2 for i, n in enumerate(sizes)]
    for (b1, step) in [(0.0, 1), (0.0, 7), (0.5, 3)]:
        dp, dg, dm, dv = ([t.cuda() for t in l] for l in (ps, gs, ms, vs))
        cp, cg, cm, cv = ([t.clone() for t in l] for l in (ps, gs, ms, vs))
        ops.adam_multi(dp, dg, dm, dv, 2e-4, b1, 0.999, 1e-6, step)
        R.adam_multi_ref(cp, cg, cm, cv, 2e-4, b1, 0.999, 1e-6, step)
        for a, b in zip(dp + dm + dv, cp + cm + cv):
            close(a, b, rtol=1e-5, atol_rel=1e-6, what=f"adam b1={b1} step={step}")
    tg, sr = [t.cuda() for t in ps], [t.cuda() for t in gs]
    ct, cs = [t.clone() for t in ps], [t.clone() for t in gs]
    ops.ema_multi(tg, sr, 0.9999)
    R.ema_multi_ref(ct, cs, 0.9999)
    for a, b in zip(tg, ct):
        close(a, b, rtol=1e-6, atol_rel=1e-7, what="ema")


# ------------------------------------------------------------------------------------------------ StyleGAN2 ops
ACT = dict(linear=1, relu=2, lrelu=3, tanh=4, sigmoid=5, elu=6, selu=7, softplus=8, swish=9)
DEF_GAIN = dict(linear=1, relu=np.sqrt(2), lrelu=np.sqrt(2), tanh=1, sigmoid=1, elu=1, selu=1, softplus=1, swish=np.sqrt(2))


@pytest.mark.parametrize("act", list(ACT))
@pytest.mark.parametrize("clamp", [-1.0, 0.7])
def test_bias_act(act, clamp):
    N, C, H, W = 3, 6, 5, 5
    n = N * C * H * W
    x, b = rnd(n, seed=1, scale=1.5), rnd(C, seed=2, scale=0.5)
    alpha, gain = 0.2, float(DEF_GAIN[act])
    y = torch.empty(n)
    (p,) = run_pair("icg_bias_act", [x, b, None, None, None, y, n, H * W, C, 0, ACT[act], alpha, gain, clamp], [5])
    close(*p, rtol=1e-5, atol_rel=1e-6, what=f"bias_act fwd {act}")
    yref = p[1].clone()
    dy = rnd(n, seed=3)
    # first-order: kernel(x=dy, xref=x, yref=y)
    (p1,) = run_pair("icg_bias_act", [dy, b, x, yref, None, torch.empty(n), n, H * W, C, 1, ACT[act], alpha, gain, clamp], [5])
    close(*p1, rtol=2e-4, atol_rel=2e-5, what=f"bias_act grad1 {act}")
    d2 = rnd(n, seed=4)
    (p2,) = run_pair("icg_bias_act", [d2, b, x, yref, dy, torch.empty(n), n, H * W, C, 2, ACT[act], alpha, gain, clamp], [5])
    close(*p2, rtol=5e-4, atol_rel=5e-5, what=f"bias_act grad2 {act}")


UPFIR = [  # N, C, H, W, fh, fw, up, down, (px0, px1, py0, py1), flip
    (2, 3, 8, 8, 4, 4, 2, 1, (2, 1, 2, 1), 0), (2, 5, 9, 7, 4, 4, 1, 2, (1, 1, 1, 1), 0),
    (1, 4, 16, 16, 4, 4, 1, 1, (1, 2, 2, 1), 1), (2, 2, 6, 6, 3, 5, 2, 2, (0, 3, -1, 2), 0),
    (1, 8, 17, 17, 4, 4, 1, 1, (-1, -1, 0, 0), 0), (1, 1, 4, 4, 1, 1, 1, 1, (0, 0, 0, 0), 0),
    (2, 3, 8, 8, 8, 8, 2, 1, (4, 3, 4, 3), 1),
]


@pytest.mark.parametrize("case", UPFIR)
def test_upfirdn2d(case):
    N, C, H, W, fh, fw, up, down, pad, flip = case
    px0, px1, py0, py1 = pad
    outW = (W * up + px0 + px1 - fw + down) // down
    outH = (H * up + py0 + py1 - fh + down) // down
    x, f = rnd(N, C, H, W, seed=1), rnd(fh, fw, seed=2)
    y = torch.empty(N, C, outH, outW)
    (p,) = run_pair("icg_upfirdn2d", [x, f, y, N, C, H, W, fh, fw, up, up, down, down, px0, px1, py0, py1, flip, 1.7,
                                      outH, outW], [2])
    close(*p, rtol=2e-5, atol_rel=2e-5, what=f"upfirdn2d {case}")


def test_sn_forward_multi_bit_identical():
    """icg_sn_forward_multi (all layers of a network in one pass) == icg_sn_forward per layer, bit for bit, including the
    in-place u / sv updates; more layers than one descriptor pack holds."""
    from ic_gan_amd import ops
    shapes = [(96, 192, 3), (3, 96, 3), (1536, 17, 1), (24, 8, 3), (768, 768, 1), (64, 128, 3), (1000, 128, 1)] * 3
    flags = [(True, False, False), (False, False, False), (True, False, False), (True, True, False), (True, False, False),
             (True, False, True), (False, False, False)] * 3
    items_a, items_b = [], []
    for i, ((co, ci, r), fl) in enumerate(zip(shapes, flags)):
        w = rnd(*((co, ci, r, r) if r > 1 else (co, ci)), seed=100 + i).cuda()
        u = rnd(1, co, seed=200 + i).cuda()
        items_a.append((w, u.clone(), torch.ones(1, device="cuda")) + fl)
        items_b.append((w, u.clone(), torch.ones(1, device="cuda")) + fl)
    many = ops.sn_prepare_many(items_a, 1e-6, True)
    for (w, u, sv, nd, up, dn), (_, u2, sv2, _, _, _), st in zip(items_b, items_a, many):
        one = ops.sn_prepare(w, u, sv, 1e-6, True, nd, up, dn)
        assert torch.equal(u, u2) and torch.equal(sv, sv2)
        for name in ("w_ohwi", "w_dgrad", "u", "v", "sigma", "w_up", "w_up_dgrad", "w_down", "w_down_dgrad"):
            a, b = getattr(one, name), getattr(st, name)
            assert (a is None) == (b is None), name
            if a is not None:
                assert torch.equal(a, b), name


@pytest.mark.parametrize("M,K,Ns", [(64, 657, (1536, 1536, 768, 768)), (16, 148, (96, 96, 96, 96)), (7, 64, (12, 40)), (128, 657, (192,))])
def test_linear_group(M, K, Ns):
    """icg_linear_group (the conditional-BN projections of a block in one launch per direction): forward and weight gradient bit for
    bit the per-layer kernels behind icg_conv2d_fprop / icg_conv2d_wgrad; the data gradient = the sum of the per-layer ones."""
    from ic_gan_amd import ops
    L = _L()
    x = rnd(M, K, seed=1).cuda()
    ws = [rnd(n, K, seed=10 + i, scale=K ** -0.5).cuda() for i, n in enumerate(Ns)]
    dys = [rnd(M, n, seed=20 + i).cuda() for i, n in enumerate(Ns)]
    outs = [torch.empty(M, n, device="cuda") for n in Ns]
    ops._linear_group(0, M, K, [(x, w, None, o, n) for w, o, n in zip(ws, outs, Ns)])
    raws = [torch.empty(K, n, device="cuda") for n in Ns]
    ops._linear_group(1, M, K, [(x, None, d, r, n) for d, r, n in zip(dys, raws, Ns)])
    wds = [w.t().contiguous() for w in ws]
    dx = torch.empty(M, K, device="cuda")
    ops._linear_group(2, M, K, [(None, wd, d, dx if i == 0 else None, n) for i, (wd, d, n) in enumerate(zip(wds, dys, Ns))])
    ref_dx = torch.zeros(M, K, dtype=torch.float64)
    for w, d, o, r, n in zip(ws, dys, outs, raws, Ns):
        one = torch.empty(M, n, device="cuda")
        L.call("icg_conv2d_fprop", x, w, None, None, one, None, None, 0, M, 1, 1, K, n, 1, 0, 1.0)
        if M >= 16:                                                 # (below 16 rows the per-layer entry takes another kernel)
            assert torch.equal(o, one)
        close(o, one, rtol=2e-5, atol_rel=2e-5, what="group fprop vs icg_conv2d_fprop")
        close(o, (x.cpu().double() @ w.cpu().double().t()).float(), rtol=2e-5, atol_rel=2e-5, what="group fprop")
        nb = L.query("icg_conv2d_wgrad_workspace_bytes", M, 1, 1, K, n, 1)
        dw = torch.empty(K, n, device="cuda")
        L.call("icg_conv2d_wgrad", x, d, dw, None, None, 0, M, 1, 1, K, n, 1, 0, torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda"), nb)
        close(r, dw, rtol=2e-5, atol_rel=2e-5, what="group wgrad vs icg_conv2d_wgrad")
        close(r, (x.cpu().double().t() @ d.cpu().double()).float(), rtol=2e-5, atol_rel=2e-5, what="group wgrad")
        ref_dx += d.cpu().double() @ w.cpu().double()
    close(dx, ref_dx.float(), rtol=3e-5, atol_rel=3e-5, what="group dgrad")


def test_sn_backward_multi_equals_per_layer():
    """icg_sn_backward_multi (the backward of ops.SNGroupFn: many layers in two launches per descriptor pack) == icg_sn_backward per
    layer, bit for bit, over all four raw-gradient forms and more layers than one pack holds."""
    from ic_gan_amd import ops
    shapes = [(96, 192, 3), (3, 96, 3), (1536, 17, 1), (24, 8, 3), (768, 768, 1), (64, 128, 3), (1000, 128, 1), (48, 384, 1)] * 3
    flags = [(True, False, False), (False, False, False), (True, False, False), (True, True, False), (True, False, False),
             (True, False, True), (False, False, False), (True, False, False)] * 3
    forms = [0, 0, 1, 2, 0, 3, 1, 1] * 3
    items = []
    for i, ((co, ci, r), fl) in enumerate(zip(shapes, flags)):
        w = rnd(*((co, ci, r, r) if r > 1 else (co, ci)), seed=100 + i).cuda()
        items.append((w, rnd(1, co, seed=200 + i).cuda(), torch.ones(1, device="cuda")) + fl)
    states = ops.sn_prepare_many(items, 1e-6, True)
    work = []
    for i, (st, form, it) in enumerate(zip(states, forms, items)):
        raw = rnd(st.raw_numel if form >= 2 else st.rows * st.cin * st.R * st.R, seed=300 + i).cuda()
        assert form < 2 or st.raw_numel == 16 * st.rows * st.cin
        work.append((raw, form, st, it[0]))
    many = ops.sn_backward_many(work)
    for (raw, form, st, like), got in zip(work, many):
        f = [None] * 4
        f[form] = raw
        one = ops._sn_backward(f[0], f[1], st, like, dw_up=f[2], dw_down=f[3])
        assert got.shape == like.shape and torch.equal(got, one), (st.rows, st.cin, st.R, form)


@pytest.mark.parametrize("rows,Cin,taps", [(32, 32, 3), (3, 96, 3), (96, 3, 3), (1000, 64, 1), (130, 70, 3), (17, 5, 1)])
def test_sn_backward_coalesced_path(rows, Cin, taps):
    """icg_sn_backward with the large scratch (tile-transposed gather) == the minimum-scratch path == the CPU reference,
    with every gradient source present at once."""
    L = _L()
    n = rows * Cin * taps * taps
    wo = rnd(n, seed=1, scale=0.1)
    uo, v, sg = rnd(rows, seed=2), rnd(Cin * taps * taps, seed=3), torch.tensor([1.7])
    hw, oh = rnd(n, seed=4), rnd(n, seed=5)
    up = rnd(16 * rows * Cin, seed=6) if taps == 3 else None
    dn = rnd(16 * rows * Cin, seed=7) if taps == 3 else None
    ref = torch.empty(rows, Cin, taps, taps)
    R.icg_sn_backward(hw, oh, up, dn, wo, uo, v, sg, rows, Cin, taps, ref, 0, torch.empty(4096, dtype=torch.uint8), 4096)
    c = lambda t: None if t is None else t.cuda()
    outs = []
    full = L.query("icg_sn_backward_scratch_bytes", rows, Cin, taps)
    for nb in (full - rows * Cin * taps * taps * 4, full):       # partial-dot slots only (direct gather) | + the transposed copy
        dw = torch.empty(rows, Cin, taps, taps, device="cuda")
        sc = torch.empty(nb, dtype=torch.uint8, device="cuda")
        L.call("icg_sn_backward", c(hw), c(oh), c(up), c(dn), c(wo), c(uo), c(v), c(sg), rows, Cin, taps, dw, 0, sc, nb)
        outs.append(dw)
        close(dw, ref, rtol=5e-5, atol_rel=5e-5, what=f"sn bwd scratch={nb}")
    close(outs[0], outs[1], rtol=1e-6, atol_rel=1e-6, what="two sn backward paths")


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, R, flags, residual(0 none / 1 same / 2 half-res), bias
    (2, 4, 4, 64, 64, 3, 0, 0, True),                      # 1 tile, K = 576 -> 2 slices
    (4, 8, 8, 128, 96, 3, PRE_AFFINE | PRE_RELU, 1, True),  # affine prologue per slice, residual in the epilogue
    (2, 16, 16, 256, 256, 3, PRE_RELU, 2, True),           # half-resolution residual in the split-K epilogue
    (3, 4, 4, 512, 40, 1, 0, 0, False),                    # 1x1: slice = channel range
    (2, 8, 8, 80, 64, 3, 0, 0, True),                      # 45 K-tiles in 2 slices: the boundary falls inside a tap sequence
    (2, 8, 8, 128, 64, 3, 0, 3, False),                    # ReLU mask applied by the split-K epilogue
])
def test_conv2d_fprop_split_k(case):
    """icg_conv2d_fprop_ws (K cut into concurrent slices + deterministic slab reduction) against the CPU reference."""
    B, H, W, Cin, Cout, R, flags, res, bias = case
    L = _L()
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs(case, 10)
    nb = L.query("icg_conv2d_fprop_workspace_bytes", B, H, W, Cin, Cout, R, rflags)
    assert nb > 0, "expected a split-K plan for this shape"
    ws = torch.empty(nb, dtype=torch.uint8)
    out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
    (pair,) = run_pair("icg_conv2d_fprop_ws", [x, w, bvec, r, out, sc, sh, ssb, B, H, W, Cin, Cout, R, rflags, 1.0, ws, nb], [4])
    close(*pair, what=f"split-K fprop {case}")
    # a too-small workspace falls back to the single-pass kernel with the same result
    out2 = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last).cuda()
    c = lambda t: None if t is None else t.cuda()
    L.call("icg_conv2d_fprop_ws", c(x), c(w), c(bvec), c(r), out2, c(sc), c(sh), ssb, B, H, W, Cin, Cout, R, rflags, 1.0,
           torch.empty(16, dtype=torch.uint8, device="cuda"), 16)
    close(out2, pair[1], what="single-pass fallback")


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, flags, residual(0 none / 1 same / 2 half-res), bias
    (2, 8, 8, 32, 32, 0, 0, True),
    (2, 16, 16, 64, 48, PRE_AFFINE | PRE_RELU, 2, True),     # GBlock conv2: BN + ReLU prologue, upsampled skip
    (1, 6, 10, 16, 40, PRE_RELU, 1, False),                 # DBlock conv1-like, ragged N, non-square
    (3, 4, 4, 256, 256, 0, 0, True),
    (2, 6, 6, 32, 48, 0, 3, False),                         # ReLU mask in the output transform
])
def test_conv2d_winograd(case):
    """Winograd F(2x2,3x3) forward (weight transform + input transform + 16 GEMMs + output transform) vs the direct conv."""
    B, H, W, Cin, Cout, flags, res, bias = case
    L = _L()
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs((B, H, W, Cin, Cout, 3, flags, res, bias), 10)
    U = torch.empty(16, Cout, Cin)
    (pu,) = run_pair("icg_wino_weight_transform", [w, U, Cout, Cin], [1]); close(*pu, what="wino U")
    nb = L.query("icg_conv2d_wino_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8)
    out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
    (pair,) = run_pair("icg_conv2d_wino_fprop", [x, U, bvec, r, out, sc, sh, ssb, B, H, W, Cin, Cout, rflags, 1.0, ws, nb], [4])
    close(*pair, rtol=1e-4, atol_rel=1e-4, what=f"winograd fprop {case}")


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, flags
    (2, 8, 8, 32, 32, 0),
    (2, 16, 16, 64, 48, PRE_AFFINE | PRE_RELU),
    (1, 6, 10, 16, 40, PRE_RELU),
    (4, 32, 32, 128, 128, 0),            # long K (tiles) -> batched split-K slabs
])
def test_conv2d_winograd_wgrad(case):
    """Winograd-domain weight gradient (input + dy transforms, 16 long-K GEMMs with split-K, G^T dU G) vs the direct one."""
    B, H, W, Cin, Cout, flags = case
    L = _L()
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs((B, H, W, Cin, Cout, 3, flags, 0, False), 20)
    dy = cl(B, Cout, H, W, seed=31)
    dw = torch.empty(9 * Cin * Cout)
    nb = L.query("icg_conv2d_wino_wgrad_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8)
    (pair,) = run_pair("icg_conv2d_wino_wgrad", [x, dy, dw, sc, sh, ssb, B, H, W, Cin, Cout, flags, ws, nb], [2])
    close(*pair, rtol=2e-4, atol_rel=2e-4, what=f"winograd wgrad {case}")


@pytest.mark.parametrize("case", [
    (2, 8, 8, 32, 32, 0),
    (2, 16, 16, 64, 48, PRE_AFFINE | PRE_RELU),
    (1, 4, 12, 16, 40, PRE_RELU),
    (4, 32, 32, 128, 128, 0),            # long K (tiles) -> batched split-K slabs
    (3, 4, 4, 64, 64, 0),                # one tile per image
])
def test_conv2d_winograd4_wgrad(case):
    """F(4x4,3x3)-domain weight gradient (36 long-K GEMMs) vs the direct one."""
    B, H, W, Cin, Cout, flags = case
    L = _L()
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs((B, H, W, Cin, Cout, 3, flags, 0, False), 20)
    dy = cl(B, Cout, H, W, seed=31)
    dw = torch.empty(9 * Cin * Cout)
    nb = L.query("icg_conv2d_wino4_wgrad_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8)
    (pair,) = run_pair("icg_conv2d_wino4_wgrad", [x, dy, dw, sc, sh, ssb, B, H, W, Cin, Cout, flags, ws, nb], [2])
    close(*pair, rtol=5e-4, atol_rel=5e-4, what=f"winograd4 wgrad {case}")


def test_gemm_tn_batched_split_k():
    L = _L()
    M, N, K, batch = 96, 64, 5000, 5
    A, Bm = rnd(batch, K, M, seed=1), rnd(batch, K, N, seed=2)
    C = torch.empty(batch, M, N)
    nb = L.query("icg_gemm_tn_batched_workspace_bytes", M, N, K, batch)
    assert nb > 16
    ws = torch.empty(nb, dtype=torch.uint8)
    (pair,) = run_pair("icg_gemm_tn_batched", [A, Bm, C, M, N, K, K * M, K * N, M * N, batch, ws, nb], [2])
    close(*pair, rtol=1e-4, atol_rel=1e-4, what="tn batched split-K")


@pytest.mark.parametrize("case", [
    # B, H, W, Cin, Cout, flags, residual(0 none / 1 same / 2 half-res), bias
    (2, 8, 8, 32, 32, 0, 0, True),
    (2, 16, 16, 64, 48, PRE_AFFINE | PRE_RELU, 2, True),
    (1, 8, 12, 16, 40, PRE_RELU, 1, False),
    (3, 4, 4, 256, 256, 0, 0, True),
    (2, 8, 12, 48, 32, 0, 3, False),                        # ReLU mask in the output transform (D data gradients)
    (16, 128, 128, 16, 24, 0, 0, True),                     # 4608 plane tiles of ONE K-tile: persistent workgroups cross an output-tile boundary every step
    (8, 100, 160, 32, 160, PRE_RELU, 1, False),             # 4536 plane tiles of two K-tiles, ragged last M tile (8000 rows) and N tile (128 + 32)
])
def test_conv2d_winograd4(case):
    """Winograd F(4x4,3x3) forward vs the direct convolution (tolerance 2e-4 of max|ref|: fp32 transforms with entries up to 8)."""
    B, H, W, Cin, Cout, flags, res, bias = case
    L = _L()
    x, w, bvec, r, sc, sh, ssb, rflags = _conv_inputs((B, H, W, Cin, Cout, 3, flags, res, bias), 10)
    U = torch.empty(36, Cout, Cin)
    (pu,) = run_pair("icg_wino4_weight_transform", [w, U, Cout, Cin], [1]); close(*pu, what="wino4 U")
    nb = L.query("icg_conv2d_wino4_workspace_bytes", B, H, W, Cin, Cout)
    ws = torch.empty(nb, dtype=torch.uint8)
    out = torch.empty(B, Cout, H, W).contiguous(memory_format=torch.channels_last)
    (pair,) = run_pair("icg_conv2d_wino4_fprop", [x, U, bvec, r, out, sc, sh, ssb, B, H, W, Cin, Cout, rflags, 1.0, ws, nb], [4])
    close(*pair, rtol=2e-4, atol_rel=2e-4, what=f"winograd4 fprop {case}")


def test_wino_weight_transform_multi_bit_identical():
    """icg_wino_weight_transform_multi (all Winograd-domain weight copies of a network in one launch) == the single-tensor
    transforms, bit for bit, for the three plane counts and more tensors than one descriptor pack holds."""
    import ctypes
    L = _L()
    shapes = [(32, 16, 36), (40, 24, 25), (96, 96, 16), (8, 260, 36), (130, 12, 25)] * 15          # 75 tensors > 64 per pack
    arr = (L.WinoWeight * len(shapes))()
    ws, us, refs = [], [], []
    fn = {16: "icg_wino_weight_transform", 25: "icg_wino4r_weight_transform", 36: "icg_wino4_weight_transform"}
    for i, (n, k, planes) in enumerate(shapes):
        w = rnd(n, 3, 3, k, seed=100 + i).cuda()
        u = torch.full((planes * n * k,), float("nan"), device="cuda")
        r = torch.empty(planes * n * k, device="cuda")
        L.call(fn[planes], w, r, n, k)
        arr[i].w, arr[i].U, arr[i].N, arr[i].K, arr[i].planes = w.data_ptr(), u.data_ptr(), n, k, planes
        ws.append(w); us.append(u); refs.append(r)
    L.call("icg_wino_weight_transform_multi", ctypes.cast(arr, ctypes.c_void_p), len(shapes))
    torch.cuda.synchronize()
    for i, (u, r) in enumerate(zip(us, refs)):
        assert torch.equal(u, r), (i, shapes[i])


# ------------------------------------------------------------------------------------------------ kNN build
@pytest.mark.parametrize("N,D,k", [(300, 64, 7), (1000, 2048, 51), (64, 16, 64), (5, 8, 3), (257, 130, 1)])
def test_knn_l2(N, D, k):
    """exact L2 top-k of every row against the table (csrc/knn.hip) against the fp64 distance matrix: the query first, k distinct
    neighbours in ascending exact distance, none closer missed -- all up to the fp32 rounding of the Gram-matrix distances
    (2e-6 absolute on unit-norm rows; two neighbours closer together than that may swap, as in faiss)."""
    L = _L()
    f = rnd(N, D, seed=3)
    f = f / f.norm(dim=1, keepdim=True)
    idx, d2 = torch.empty(N, k, dtype=torch.int64, device="cuda"), torch.empty(N, k, device="cuda")
    nb = L.query("icg_knn_l2_workspace_bytes", N, D)
    L.call("icg_knn_l2", f.cuda(), N, D, k, idx, d2, torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda"), nb)
    idx, d2 = idx.cpu(), d2.cpu().double()
    f64 = f.double()
    sq = (f64 * f64).sum(1)
    dist = (sq[:, None] + sq[None, :] - 2.0 * (f64 @ f64.t())).clamp_(min=0)
    dist[torch.arange(N), torch.arange(N)] = -1.0
    tol = 2e-6
    assert torch.equal(idx[:, 0], torch.arange(N)), "the query row must rank first"
    assert all(len(set(r.tolist())) == k for r in idx), "duplicate neighbours"
    dg = dist.gather(1, idx)
    assert bool((dg[:, 1:] >= dg[:, :-1] - tol).all()), "neighbours not in ascending distance"
    kth = dist.sort(1).values[:, k - 1]
    assert bool((dg[:, -1] <= kth + tol).all()), "a closer row was missed"
    assert float((d2 - dg.clamp(min=0)).abs().max()) <= tol, float((d2 - dg.clamp(min=0)).abs().max())


def test_knn_l2_ties_and_duplicates():
    """identical rows: mutual distances are 0 up to the fp32 rounding of |a|^2 + |b|^2 - 2 a.b (as in faiss) -> the query itself
    is forced first, the rest of the duplicate group follows at ~0, rows stay sorted"""
    L = _L()
    N, D, k = 40, 32, 6
    f = rnd(N, D, seed=5)
    f[3:11] = f[3]
    f = f.cuda()
    idx, d2 = torch.empty(N, k, dtype=torch.int64, device="cuda"), torch.empty(N, k, device="cuda")
    nb = L.query("icg_knn_l2_workspace_bytes", N, D)
    L.call("icg_knn_l2", f, N, D, k, idx, d2, torch.empty(nb, dtype=torch.uint8, device="cuda"), nb)
    idx, d2 = idx.cpu(), d2.cpu()
    assert all(int(idx[i, 0]) == i for i in range(N))
    for q in range(3, 11):
        got = idx[q, 1:].tolist()
        assert len(set(got)) == k - 1 and all(3 <= j < 11 and j != q for j in got), (q, idx[q].tolist())
        assert float(d2[q, 1:].abs().max()) < 1e-5
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())


# ------------------------------------------------------------------------------------------------ fp16 gather convolution
HCONV_CASES = [
    # B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad, zero_insert
    (2, 16, 16, 64, 16, 16, 64, 3, 1, 1, 0),             # 3x3 'same', 64-column tile
    (3, 9, 7, 32, 9, 7, 128, 3, 1, 1, 0),                # odd sizes, ragged last pixel tile, a single channel slice, 128-column tile
    (2, 8, 8, 96, 8, 8, 192, 1, 1, 0, 0),                # 1x1, 96-column tile
    (1, 4, 4, 32, 4, 4, 64, 3, 1, 1, 0),                 # 16 pixels: one partial tile
    (2, 17, 17, 64, 8, 8, 128, 3, 2, 0, 0),              # stride 2 (the discriminator's down-sampling conv after its blur)
    (2, 16, 16, 64, 8, 8, 64, 1, 2, 0, 0),               # 1x1 stride 2
    (2, 8, 8, 128, 17, 17, 64, 3, 1, 2, 2),              # stride-2 transposed 3x3 (zero-inserted source, extent 15, output 17)
    (2, 8, 8, 128, 17, 17, 64, 3, 1, 0, 2),              # data gradient geometry of a padded stride-2 convolution (adjoint pad)
    (2, 8, 8, 64, 16, 16, 96, 3, 1, 1, 2),               # output grid past the zero-inserted source (those positions read zeros)
    (2, 12, 12, 64, 10, 10, 64, 5, 1, 1, 0),             # 5x5, partial padding
    (2, 8, 8, 64, 15, 15, 64, 1, 1, 0, 2),               # 1x1 over a zero-inserted source: three of the four phases have no tap at all
    (1, 6, 5, 32, 13, 11, 64, 3, 1, 1, 2),               # odd phase grids (13 x 11 outputs: 7/6 rows, 6/5 columns per phase), pad 1
    (2, 7, 7, 64, 16, 16, 128, 4, 1, 2, 2),              # 4x4 taps: two per phase and axis
    (4, 32, 32, 512, 32, 32, 512, 3, 1, 1, 0),           # cfg4, resolution 32: 144 K-tiles
    (3, 64, 64, 256, 129, 129, 128, 3, 1, 2, 2),         # cfg4 synthesis b128.conv0 (up): 50k output pixels
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", HCONV_CASES)
def test_conv2d_g_fprop_f16(case):
    """icg_conv2d_g_fprop_f16 (csrc/hconv.hip): fp16 operands, fp32 accumulation, one rounding -- against the same contraction in
    fp64 rounded to fp16 once (oracle/kernel_ref.py).  Tolerance: 2 fp16 ulps of the value + 1e-3 of the tensor maximum (an fp32
    sum that lands next to a rounding boundary rounds the other way than the fp64 sum)."""
    B, H, W, Cin, Ho, Wo, Cout, R, stride, pad, zins = case
    assert _hconv_applies(Cin, Cout, R, stride, zins)
    x = cl(B, Cin, H, W, seed=1).half()
    w = (rnd(Cout, R, R, Cin, seed=2) * (Cin * R * R) ** -0.5).half()
    out = torch.zeros(B, Cout, Ho, Wo, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    (p,) = run_pair("icg_conv2d_g_fprop_f16", [x, w, out, B, H, W, Cin, Ho, Wo, Cout, R, stride, pad, zins], [2])
    close(*p, rtol=2e-3, atol_rel=1e-3, what="fp16 gather conv %r" % (case,))


def _hconv_applies(Cin, Cout, taps, stride, zins):
    return _L().query("icg_conv2d_g_fprop_f16_applies", Cin, Cout, taps, stride, zins) == 1 and \
        R.icg_conv2d_g_fprop_f16_applies(Cin, Cout, taps, stride, zins) == 1


HWGRAD_CASES = [
    # B, Hin, Win, Cin, Hout, Wout, Cout, R, stride, pad
    (2, 16, 16, 64, 16, 16, 64, 3, 1, 1),                # 3x3 'same': 576 (tap, ci) rows = 4.5 row tiles, 64 of 128 columns
    (3, 9, 7, 32, 9, 7, 128, 3, 1, 1),                   # odd sizes: every 32-pixel block has a zero-filled tail
    (2, 40, 40, 32, 40, 40, 32, 1, 1, 0),                # 1x1; 40-pixel rows: a full and a partial block per row
    (2, 17, 17, 64, 8, 8, 128, 3, 2, 0),                 # stride 2
    (2, 17, 17, 128, 8, 8, 64, 3, 2, 2),                 # the zero-inserted direction's geometry (roles of x and dy swapped by the caller)
    (2, 12, 12, 64, 10, 10, 96, 5, 1, 1),                # 5x5, partial padding, 96 output channels
    (4, 32, 32, 512, 32, 32, 512, 3, 1, 1),              # cfg4, resolution 32: 36 x 4 tiles, 8 pixel slices
    (2, 128, 128, 128, 128, 128, 128, 3, 1, 1),          # cfg4, resolution 128: 9 row tiles, 100+ slices
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", HWGRAD_CASES)
def test_conv2d_g_wgrad_f16(case):
    """icg_conv2d_g_wgrad_f16 (csrc/hwgrad.hip): fp16 operands through the transposing LDS read into v_mfma_f32_16x16x32_f16, fp32
    accumulation over pixel slices, fp32 result -- against the fp32 reference of the same contraction on the same fp16 values.
    Tolerance: 5e-5 of the tensor maximum + 1e-4 relative (fp32 sums of up to 32k exact products in a different order)."""
    B, H, W, Cin, Ho, Wo, Cout, taps, stride, pad = case
    L = _L()
    assert L.query("icg_conv2d_g_wgrad_f16_applies", Cin, Cout, taps, stride) == 1 and R.icg_conv2d_g_wgrad_f16_applies(Cin, Cout, taps, stride) == 1
    x = cl(B, Cin, H, W, seed=1).half()
    dy = cl(B, Cout, Ho, Wo, seed=2).half()
    dw = torch.zeros(taps, taps, Cin, Cout)
    nb = L.query("icg_conv2d_g_wgrad_f16_workspace_bytes", B, Ho, Wo, Cin, Cout, taps)
    ws = torch.zeros(nb, dtype=torch.uint8)
    (p,) = run_pair("icg_conv2d_g_wgrad_f16", [x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, taps, stride, pad, ws, nb], [2])
    close(*p, rtol=1e-4, atol_rel=5e-5, what="fp16 weight gradient %r" % (case,))


# ------------------------------------------------------------------------------------------------ attention scores + softmax
ATTN_CASES = [
    # B, n, m, d
    (2, 64, 128, 48), (3, 32, 256, 24), (1, 96, 1024, 48), (2, 128, 1024, 24), (2, 64, 512, 16), (1, 32, 128, 8), (2, 64, 384, 64),
    (1, 4096, 1024, 48),          # the generator's attention block at 64 x 64 (one image)
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ATTN_CASES)
def test_attn_scores_softmax(case):
    """icg_attn_scores_softmax (csrc/attn.hip) against softmax(bmm(theta, phi^T)) in fp32 on the CPU; scores of a few units, so the
    probabilities span many decades: 2e-5 of the row maximum + 1e-4 relative."""
    B, n, m, d = case
    assert _L().query("icg_attn_scores_softmax_applies", n, m, d) == 1 and R.icg_attn_scores_softmax_applies(n, m, d) == 1
    theta = rnd(B, n, d, seed=1)
    phi = rnd(B, m, d, seed=2, scale=0.7)
    beta = torch.zeros(B, n, m)
    (p,) = run_pair("icg_attn_scores_softmax", [theta, phi, beta, B, n, m, d], [2])
    close(*p, rtol=1e-4, atol_rel=2e-5, what="attention scores + softmax %r" % (case,))
    assert float((p[0].cpu().sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 64, 128, 96), (1, 96, 1024, 192), (3, 32, 512, 96), (2, 128, 1024, 96), (1, 4096, 1024, 192)])
def test_attn_dscores(case):
    """icg_attn_dscores (csrc/attn.hip: dP = dO V^T with the softmax backward in the epilogue) against bmm + the softmax backward
    formula in fp32 on the CPU; beta from a softmax of scores of a few units (probabilities over many decades)."""
    B, n, m, dv = case
    assert _L().query("icg_attn_dscores_applies", n, m, dv) == 1 and R.icg_attn_dscores_applies(n, m, dv) == 1
    assert _L().query("icg_attn_dscores_applies", n, m, 48) == 0
    do = rnd(B, n, dv, seed=1)
    v = rnd(B, m, dv, seed=2, scale=0.5)
    beta = torch.softmax(rnd(B, n, m, seed=3, scale=2.0), -1).contiguous()
    ds = torch.zeros(B, n, m)
    (p,) = run_pair("icg_attn_dscores", [do, v, beta, ds, B, n, m, dv], [3])
    close(*p, rtol=1e-4, atol_rel=2e-5, what="attention dS %r" % (case,))
    assert float(p[0].cpu().sum(-1).abs().max()) < 1e-4 * float(p[1].abs().max()) * m ** 0.5 + 1e-5      # rows of dS sum to zero


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(2, 8, 8, 4, 16), (1, 64, 64, 48, 192), (3, 16, 32, 24, 96), (2, 6, 10, 8, 4)])
def test_attn_split_pool(case):
    """icg_attn_split_pool / _bwd (csrc/attn.hip) against slicing + F.max_pool2d and its autograd on the CPU: pure data movement and
    comparisons, so bit for bit; windows with tied maxima route to the first one in both."""
    B, H, W, d, dv = case
    C = 2 * d + dv
    y = rnd(B, H, W, C, seed=1)
    y[0, :2, :2, d:d + 4] = 0.25                                   # a window of ties in phi
    y[-1, 2:4, 2:4, 2 * d:2 * d + 4] = -1.0                        # ... and in g
    th, ph, g = torch.zeros(B, H * W, d), torch.zeros(B, H * W // 4, d), torch.zeros(B, H * W // 4, dv)
    pairs = run_pair("icg_attn_split_pool", [y, th, ph, g, B, H, W, d, dv], [1, 2, 3])
    for got, ref in pairs:
        assert torch.equal(got.cpu(), ref)
    dth, dph, dg = rnd(B, H * W, d, seed=2), rnd(B, H * W // 4, d, seed=3), rnd(B, H * W // 4, dv, seed=4)
    dy = torch.zeros(B, H, W, C)
    (p,) = run_pair("icg_attn_split_pool_bwd", [y, dth, dph, dg, dy, B, H, W, d, dv], [4])
    assert torch.equal(p[0].cpu(), p[1])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [7, 1024, 192 * 384, 300001])
def test_attn_gamma_fold(n):
    """icg_attn_gamma_scale / icg_attn_gamma_bwd (csrc/attn.hip: gamma of the attention block folded into its output projection)."""
    gamma = torch.tensor([0.37])
    wa, wb = rnd(n, seed=1), rnd(n, seed=2)
    pairs = run_pair("icg_attn_gamma_scale", [gamma, wa, torch.zeros(n), wb, torch.zeros(n), n], [2, 4])
    for got, ref in pairs:
        assert torch.equal(got.cpu(), ref)
    (p,) = run_pair("icg_attn_gamma_scale", [gamma, wa, torch.zeros(n), None, None, n], [2])
    assert torch.equal(p[0].cpu(), p[1])
    dws = rnd(n, seed=3)
    pairs = run_pair("icg_attn_gamma_bwd", [gamma, dws, wa, torch.zeros(n), torch.zeros(1), n], [3, 4])
    assert torch.equal(pairs[0][0].cpu(), pairs[0][1])
    close(*pairs[1], rtol=1e-6, atol_rel=1e-6, what="dgamma")


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(1000, 64), (37, 8), (5000, 512), (70000, 128), (300, 2048), (4 * 129 * 129, 256)])
def test_colsum_f16(rows, C):
    """icg_colsum_f16 (bias gradient of bias_act in StyleGAN2's fp16 blocks): fp32 column sums of an fp16 [rows][C] tensor against fp64."""
    L = _L()
    assert L.query("icg_colsum_f16_applies", C) == 1 and R.icg_colsum_f16_applies(C) == 1
    assert L.query("icg_colsum_f16_applies", 24) == 0 and L.query("icg_colsum_f16_applies", 4) == 0
    x = rnd(rows, C, seed=1).half()
    nb = L.query("icg_colsum_f16_workspace_bytes", rows, C)
    (p,) = run_pair("icg_colsum_f16", [x, rows, C, torch.zeros(C), torch.zeros(nb, dtype=torch.uint8), nb], [3])
    close(*p, rtol=1e-5, atol_rel=1e-5, what="fp16 column sums %d x %d" % (rows, C))

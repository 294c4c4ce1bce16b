"""SURVEY 8(f) N1, second order, ORACLE-ANCHORED: one SynthesisLayer / ToRGBLayer differentiated twice the way the
path-length regulariser does (training/loss.py:120-139), against float64 results of the UNMODIFIED reference's layer classes
(tests/golden/make_golden_sg2_layers2.py -> tests/golden/sg2_layers_second_order.npz).  Both routes of this repository are held to the
fixture -- the composed, arbitrarily differentiable operators and the hand-written second-order nodes of stylegan_ops/fused_layers.py --

  * on the CPU with the kernels emulated by oracle/kernel_ref.py (host logic of the adjoints; runs in the -m "not gpu" suite);
  * on the MI355X through the C-ABI: fp32 activations at rel <= 1e-5 of max|ref| per tensor, fp16 activations at the stated bounds.

The cases keep every activation >= 5e-6 away from the kinks of lrelu / clamp (fixture array `kink_margin`), so that an fp32
implementation cannot land on the other side of one (profiles/r06_sg2_nondeterminism.txt)."""
import os

import numpy as np
import pytest
import torch

from tests.stylegan_cases import SG2_LAYER2, sg2_layer2_run

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sg2_layers_second_order.npz")


def _layer(case):
    from ic_gan_amd.stylegan2 import networks as N
    kind, cin, cout, wd, res, up, noise_mode, clamp, n = case
    if kind == "synthesis":
        return N.SynthesisLayer(cin, cout, w_dim=wd, resolution=res, up=up, conv_clamp=clamp)
    return N.ToRGBLayer(cin, cout, w_dim=wd, conv_clamp=clamp)


def _torgb(layer, x, w, img):
    return layer(x, w, fused_modconv=False, img=img)          # (this repository's layer accumulates the image itself)


def _compare(idx, got, tol, tol_first, what, tol_y=None):
    gold = np.load(GOLD)
    keys = sorted(k.split("/", 1)[1] for k in gold.files if k.startswith("%d/" % idx) and not k.endswith("kink_margin"))
    assert float(gold["%d/kink_margin" % idx]) > 5e-6
    assert set(keys) <= set(got) | {"dd_p/noise_strength", "dd_p/bias"}, (sorted(got), keys)
    worst, bad = {}, []
    for k in keys:
        if k not in got:
            continue
        ref = gold["%d/%s" % (idx, k)].astype(np.float64)
        scale = float(np.abs(ref).max())
        if scale == 0.0:
            assert float(np.abs(got[k]).max()) == 0.0, k
            continue
        err = float(np.abs(got[k] - ref).max()) / scale
        worst[k] = err
        bound = tol_first if k in ("y", "g", "gx") else tol
        if k == "y" and tol_y is not None:
            bound = tol_y
        if not (np.isfinite(got[k]).all() and err <= bound):
            bad.append("%s %.3e > %.1e" % (k, err, bound))
    if os.environ.get("ICG_REPORT"):          # measurement record (profiles/r06_sg2_layers2_errors.txt)
        with open(os.environ["ICG_REPORT"], "a") as f:
            f.write("%-20s case %d %-60s %s\n" % (what, idx, str(SG2_LAYER2[idx]), " ".join("%s=%.2e" % kv for kv in sorted(worst.items()))))
    assert not bad, "%s case %d: %s | all: %s" % (what, idx, "; ".join(bad), {k: "%.2e" % v for k, v in worst.items()})
    return worst


@pytest.fixture
def emu(monkeypatch):
    from oracle import kernel_ref
    kernel_ref.install(monkeypatch)


@pytest.mark.parametrize("idx", range(len(SG2_LAYER2)))
@pytest.mark.parametrize("fused", [False, True])
def test_second_order_layer_vs_reference_float64_cpu_emulated(idx, fused, emu):
    from ic_gan_amd import _lib as L
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    seen, orig = [], L.call
    L.call = lambda name, *a: (seen.append(name), orig(name, *a))[1]
    try:
        got = sg2_layer2_run(idx, SG2_LAYER2[idx], _layer, dtype=torch.float32, torgb_call=_torgb, context=FL.second_order if fused else None)
    finally:
        L.call = orig
    assert bool({"icg_sg2_act_bwd2", "icg_sg2_torgb_bwd2"} & set(seen)) == fused, sorted(set(seen))
    _compare(idx, got, 1e-5, 1e-5, "emulated fused" if fused else "emulated composed")


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(SG2_LAYER2)))
@pytest.mark.parametrize("fused", [False, True])
def test_second_order_layer_vs_reference_float64_hip_fp32(idx, fused):
    from ic_gan_amd import _lib as L
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    seen, orig = [], L.call
    L.call = lambda name, *a: (seen.append(name), orig(name, *a))[1]
    try:
        got = sg2_layer2_run(idx, SG2_LAYER2[idx], _layer, dtype=torch.float32, device="cuda", torgb_call=_torgb,
                             context=FL.second_order if fused else None)
    finally:
        L.call = orig
    second = {"icg_sg2_act_bwd2", "icg_sg2_torgb_bwd2"} & set(seen)
    assert bool(second) == fused, sorted(set(seen))
    # measured (profiles/r06_sg2_layers2_errors.txt): every tensor of every case <= 3.8e-6, most <= 1e-6.  One exception by construction:
    # the forward `y` of the case whose clamp bites (conv_clamp = 0.9) is compared on the scale of the CLAMPED output (max 0.9) while the
    # 128-channel 3x3 layer behind it -- on the F(4x4,3x3) Winograd route -- carries round-off on the scale of the unclamped values (~5):
    # 1.55e-5 of 0.9 on both the composed and the fused route; bound 3e-5 for that one array
    clamp = SG2_LAYER2[idx][7]
    _compare(idx, got, 1e-5, 1e-5, "HIP fused" if fused else "HIP composed", tol_y=3e-5 if (clamp is not None and clamp < 1) else None)


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(len(SG2_LAYER2)))
@pytest.mark.parametrize("fused", [False, True])
def test_second_order_layer_vs_reference_float64_hip_fp16(idx, fused):
    """fp16 activations (the num_fp16_res blocks): x and every activation tensor round to fp16, parameters and styles stay fp32.  A SANITY bound,
    not a precision claim: at fp16 resolution (4.9e-4) some pre-activations land on the other side of the lrelu kink than in the float64
    reference, and each such element moves second-order results by ~1e-2 of the tensor maximum.  Measured (profiles/r06_sg2_layers2_errors.txt):
    forward <= 1e-3; first-order <= 3.3e-2; second-order <= 1.2e-1 in the two lrelu cases with flips, <= 1.5e-3 in the cases without -- and
    the composed and the fused route agree with EACH OTHER to three digits of those errors (8.70e-2 / 8.71e-2, 1.21e-1 / 1.21e-1): the
    noise is the layer's, not an implementation's.  The fused-vs-composed fp16 comparison proper is tests/test_sg2_fused_gpu.py."""
    from ic_gan_amd.stylegan_ops import fused_layers as FL
    if SG2_LAYER2[idx][7] is not None and SG2_LAYER2[idx][7] < 1:
        pytest.skip("a clamp at O(1): fp16 rounding moves activations across it (no kink margin at fp16 resolution)")
    got = sg2_layer2_run(idx, SG2_LAYER2[idx], _layer, dtype=torch.float32, device="cuda", act_dtype=torch.float16, torgb_call=_torgb,
                         context=FL.second_order if fused else None)
    _compare(idx, got, 2.5e-1, 7e-2, "HIP fp16 fused" if fused else "HIP fp16 composed", tol_y=3e-3)

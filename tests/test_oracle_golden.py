"""Pins oracle/biggan_oracle.py to outputs of the unmodified reference
(tests/golden/*.npz, produced by tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import biggan_oracle as O
from oracle import synth
from tests.helpers import CASES, REAL_CASES, check_group, load_golden


def _fresh(g):
    return synth.synth_state(g["gspec"], seed=11), synth.synth_state(g["dspec"], seed=22)


@pytest.mark.parametrize("case", CASES + REAL_CASES)
def test_oracle_forward_matches_reference(case):
    g = load_golden(case)
    cfg = g["cfg"]
    gsd, dsd = _fresh(g)
    dims = O.g_dims(cfg)
    assert dims["dim_z"] == int(g["dim_z"])
    gb = int(g["g_batch"])
    c = synth.CondSampler(cfg, dims["dim_z"], gb, seed=5)()
    z = c[0] if isinstance(c, tuple) else c
    lab = c[1] if cfg["class_cond"] else None
    fg = c[-1] if cfg["instance_cond"] else None
    taps = {}
    with torch.no_grad():
        img = O.generator_forward(gsd, cfg, z, lab, fg, True, taps)
        logit = O.discriminator_forward(dsd, cfg, img, lab, fg, True, taps)
    if "fwd/img" in g:
        np.testing.assert_allclose(img.numpy(), g["fwd/img"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logit.numpy(), g["fwd/logit"], rtol=1e-4, atol=1e-4)
    taps["img"] = img
    check_group(g, "fwd/taps/", taps, rtol=1e-4, atol=1e-6, what="tap ")
    check_group(g, "fwd/G_state/", gsd, rtol=1e-5, atol=1e-7, what="G buf ")
    check_group(g, "fwd/D_state/", dsd, rtol=1e-5, atol=1e-7, what="D buf ")


@pytest.mark.parametrize("case", CASES + REAL_CASES)
def test_oracle_train_step_matches_reference(case):
    g = load_golden(case)
    cfg = g["cfg"]
    gsd, dsd = _fresh(g)
    ema_sd = {k: v.clone() for k, v in gsd.items()}
    dims = O.g_dims(cfg)
    gb, steps = int(g["g_batch"]), int(g["steps"])
    opt_g = O.AdamState(O.param_names(gsd), cfg["G_lr"], cfg["G_B1"], cfg["G_B2"], cfg["adam_eps"])
    opt_d = O.AdamState(O.param_names(dsd), cfg["D_lr"], cfg["D_B1"], cfg["D_B2"], cfg["adam_eps"])
    samp = synth.CondSampler(cfg, dims["dim_z"], gb, seed=7)
    dbatch = gb * cfg["num_D_accumulations"] * cfg["num_D_steps"]
    for s in range(steps):
        x, y, f = synth.synth_batch(cfg, dbatch, seed=100 + s)
        m, gg, dg = O.train_step(gsd, dsd, ema_sd, cfg, opt_g, opt_d, x, y, f, samp, s + 1, gb)
        ref = g["losses"][s]
        np.testing.assert_allclose([m["G_loss"], m["D_loss_real"], m["D_loss_fake"]], ref, rtol=2e-4, atol=2e-4)
        if s == 0:
            check_group(g, "step1/G_grad/", {k: v for k, v in gg.items() if v is not None}, 2e-3, 1e-7, "G grad ")
            check_group(g, "step1/D_grad/", {k: v for k, v in dg.items() if v is not None}, 2e-3, 1e-7, "D grad ")
        check_group(g, f"step{s + 1}/G_state/", gsd, 2e-3, 1e-6, "G ")
        check_group(g, f"step{s + 1}/D_state/", dsd, 2e-3, 1e-6, "D ")
        check_group(g, f"step{s + 1}/EMA_state/", ema_sd, 2e-3, 1e-6, "EMA ")
